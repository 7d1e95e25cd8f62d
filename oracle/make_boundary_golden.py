"""Generates tests/golden/boundary_golden.json by running the REFERENCE engine
(oracle/_ref/execute_engine, built from /root/reference by oracle/build_ref.sh) in dry-run
mode on probe plugins written for this purpose.  The fixture holds the probe templates, the
call parameters and the command lines the reference engine logged - data, not reference source.

Run from the repo root:  python oracle/make_boundary_golden.py
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENGINE = os.path.join(ROOT, "oracle", "_ref", "execute_engine")

PROBES = [
    {
        "id": "probe_vars",
        "commands": ["tool --in $input --out ${output}.bin $dir/sub/$leaf \"$literal kept\" end"],
        "inputs": {"input": ("file", "raw", None), "dir": ("text", None, "/base"), "leaf": ("text", None, "leaf.txt"),
                   "literal": ("text", None, "unused")},
        "outputs": {"output": ("file", "raw")},
        "calls": [{"input": "/x/in.raw", "output": "/x/out"},
                  {"input": "/x/in.raw", "output": "/x/out", "dir": "/other", "leaf": "z"}],
    },
    {
        "id": "probe_cond",
        "commands": ["tool [$flag ? --flag] [$mode == fast ? --fast : --slow] [$mode != fast ? --careful $level ] $input",
                     "second --n $level"],
        "inputs": {"input": ("file", "raw", None), "flag": ("bool", None, "false"), "mode": ("text", None, "slow"),
                   "level": ("int", None, "3")},
        "outputs": {},
        "calls": [{"input": "/i"}, {"input": "/i", "flag": "true", "mode": "fast"},
                  {"input": "/i", "flag": "yes", "mode": "fast", "level": "9"}],
    },
]


def plugin_xml(p):
    lines = ['<plugin id="%s" name="%s" version="1">' % (p["id"], p["id"]), "<description>probe</description>", "<commands>"]
    for c in p["commands"]:
        lines.append("<command>%s</command>" % c.replace("&", "&amp;").replace('"', "&quot;").replace(">", "&gt;"))
    lines.append("</commands><inputs>")
    for name, (typ, fmt, val) in p["inputs"].items():
        attrs = 'name="%s" type="%s"' % (name, typ)
        if fmt:
            attrs += ' format="%s"' % fmt
        if val is not None:
            attrs += ' value="%s"' % val
        lines.append('<param %s label="%s"/>' % (attrs, name))
    lines.append("</inputs><outputs>")
    for name, (typ, fmt) in p["outputs"].items():
        lines.append('<param name="%s" type="%s" format="%s" label="%s"/>' % (name, typ, fmt, name))
    lines.append("</outputs></plugin>")
    return "\n".join(lines)


REFERENCE_ROOT = "/root/reference/engine/test"   # the reference's own plugin / workflow repository
ALIGN_EXAMPLE = os.path.join(ROOT, "oracle", "_ref", "align_workflow")

# Dry runs of what the reference itself holds: its fastr plugin (engine/test/plugins/fastr/FastR.xml:1-23),
# its bwa_pe_sam workflow (engine/test/workflows/bwa_pe_sam.xml) and its C++ example
# (engine/examples/c++/align_workflow.cpp:23-73, compiled unmodified).  Only call parameters and the
# command lines the reference engine logged are stored; the descriptors stay in the reference tree.
REFERENCE_CALLS = [
    {"kind": "plugin", "id": "fastr",
     "params": {"input_bam": "/d/in.bam", "input_interval": "/d/in.intervals", "ref_file": "/d/hg19.fa",
                "output_bam": "/d/out.bam", "use_sw": "true"}},
    {"kind": "plugin", "id": "fastr",
     "params": {"input_bam": "/d/in.bam", "input_interval": "/d/in.intervals", "ref_file": "/d/hg19.fa",
                "output_bam": "/d/out.bam", "num_threads": "16"}},
    {"kind": "workflow", "id": "bwa_pe_sam",
     "params": {"input-read1": "/d/r1.fq", "input-read2": "/d/r2.fq", "ref_genome": "/d/hg19", "output_sam": "/d/out.sam"}},
    {"kind": "example", "id": "align_workflow", "params": {}},
]


def _parse_log(log, root):
    """(workdir, commands, moves) from the engine's dry-run log; the mkdtemp directory reads <WORK>."""
    work = re.findall(r"Changing working directory to (\S+)", log)
    workdir = work[0] if work else ""
    norm = lambda t: t.replace(workdir, "<WORK>").replace(root, "<ROOT>") if workdir else t.replace(root, "<ROOT>")  # noqa: E731
    commands = [norm(c) for c in re.findall(r"Executing: (.*)", log)]
    moves = [[norm(a), norm(b)] for a, b in re.findall(r"Moving file from (\S+) to (\S+)", log)]
    return commands, moves


def reference_fixtures():
    """Runs the reference engine on a scratch copy of the reference's repository (the engine creates its
    workspace inside the repository it is given, and the reference tree is read-only)."""
    if not os.path.isdir(REFERENCE_ROOT):
        return None
    out = []
    tmp = tempfile.mkdtemp()
    try:
        root = os.path.join(tmp, "oge")
        shutil.copytree(REFERENCE_ROOT, root)
        for call in REFERENCE_CALLS:
            if call["kind"] == "example":
                argv = [ALIGN_EXAMPLE, root]
            else:
                argv = [ENGINE, root, "--test", call["kind"], call["id"]]
                for k, v in call["params"].items():
                    argv += ["--" + k, v]
            log = subprocess.run(argv, capture_output=True, text=True, cwd=root).stdout
            commands, moves = _parse_log(log, root)
            out.append(dict(call, commands=commands, moves=moves))
    finally:
        shutil.rmtree(tmp)
    return out


def main():
    if not os.path.exists(ENGINE):
        sys.exit("build the reference engine first: bash oracle/build_ref.sh")
    ref = reference_fixtures()
    if ref is not None:
        path = os.path.join(ROOT, "tests", "golden", "boundary_reference_golden.json")
        with open(path, "w") as f:
            json.dump(ref, f, indent=1)
        print("wrote", path, len(ref), "calls")
    out = []
    tmp = tempfile.mkdtemp()
    try:
        os.makedirs(os.path.join(tmp, "plugins"))
        with open(os.path.join(tmp, "ctc.ini"), "w") as f:
            f.write("verbose = yes\ndebug = yes\nwork_dir = workspace\nplugin_dir = plugins\nworkflow_dir = workflows\nlog_file =\n")
        with open(os.path.join(tmp, "catalog.xml"), "w") as f:
            f.write('<suites><suite label="probes" id="probes">' +
                    "".join('<plugin id="%s" file="plugins/%s.xml"/>' % (p["id"], p["id"]) for p in PROBES) +
                    "</suite></suites>")
        for p in PROBES:
            xml = plugin_xml(p)
            with open(os.path.join(tmp, "plugins", p["id"] + ".xml"), "w") as f:
                f.write(xml)
            for call in p["calls"]:
                argv = [ENGINE, tmp, "--test", "plugin", p["id"]]
                for k, v in call.items():
                    argv += ["--" + k, v]
                log = subprocess.run(argv, capture_output=True, text=True, cwd=tmp).stdout
                cmds = re.findall(r'Executing: "(.*)"', log)
                out.append({"plugin": p["id"], "xml": xml, "params": call,
                            "commands": [c.replace(tmp, "<ROOT>") for c in cmds]})
    finally:
        shutil.rmtree(tmp)
    path = os.path.join(ROOT, "tests", "golden", "boundary_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(out), "calls")


if __name__ == "__main__":
    main()
