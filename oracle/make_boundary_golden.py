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


def main():
    if not os.path.exists(ENGINE):
        sys.exit("build the reference engine first: bash oracle/build_ref.sh")
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
