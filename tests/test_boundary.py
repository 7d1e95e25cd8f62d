"""The drop-in boundary: the unchanged CTC engine expands a plugin template and system()s
`ctc_contract`.  CPU tests pin the host-side mirror of the template language against command
lines logged by the reference engine itself (tests/golden/boundary_golden.json, made by
oracle/make_boundary_golden.py) and, when oracle/_ref/execute_engine is present, against the
reference engine live.  The GPU test runs BASELINE config 1 for real through the engine."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from ctc_amd import engine_boundary as eb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENGINE = os.path.join(ROOT, "oracle", "_ref", "execute_engine")
TOOL = os.path.join(ROOT, "ctc_amd", "bin", "ctc_contract")


def _golden():
    with open(os.path.join(ROOT, "tests", "golden", "boundary_golden.json")) as f:
        return json.load(f)


def test_template_language_matches_reference_engine_vectors(tmp_path):
    for i, case in enumerate(_golden()):
        path = tmp_path / ("p%d.xml" % i)
        path.write_text(case["xml"])
        got = eb.Plugin(str(path)).exec_commands(case["params"], {"ENGINE_DIR": "<ROOT>/"})
        assert got == case["commands"], case["plugin"]


def test_template_errors():
    with pytest.raises(eb.EngineError):
        eb.resolve_word("$nope_not_defined_anywhere", {}, {}, env={})
    with pytest.raises(eb.EngineError):
        eb.resolve_stream("tool [$a == b --x", {"a": "b"}, {})
    with pytest.raises(eb.EngineError):
        eb.plugin_commands("contract", {"not_a_parameter": "1"})
    assert eb.resolve_word("$HOME_X/bin", {}, {}, env={"HOME_X": "/h"}) == "/h/bin"


def test_contract_plugin_command_line():
    cmds = eb.plugin_commands("contract", dict(spec="adc,db=abc", extents="a=8,b=6,c=5,d=7",
                                               input_a="/t/A.bin", input_b="/t/B.bin", output_c="/t/C.bin"))
    assert len(cmds) == 1
    words = cmds[0].split()
    assert words[0].endswith("plugins/contract/../../../bin/ctc_contract")
    assert os.path.realpath(words[0]) == os.path.realpath(TOOL)
    assert words[1:] == ["--spec", "adc,db=abc", "--extents", "a=8,b=6,c=5,d=7", "--dtype", "f64", "--a", "/t/A.bin",
                         "--b", "/t/B.bin", "--out", "/t/C.bin", "--alpha", "1", "--beta", "0", "--device", "0"]
    cmds = eb.plugin_commands("contract", dict(spec="s", extents="e", input_a="a", input_b="b", output_c="c",
                                               shard="a:1/4", dtype="f32"))
    assert cmds[0].split()[-2:] == ["--shard", "a:1/4"] and "--dtype f32" in cmds[0]
    cmds = eb.plugin_commands("contract", dict(spec="s", extents="e", input_a="a", input_b="b", output_c="c", ngpu="8"))
    assert cmds[0].split()[-2:] == ["--ngpu", "8"] and "--shard" not in cmds[0]
    cmds = eb.plugin_commands("contract", dict(spec="s", extents="e", input_a="a", input_b="b", output_c="c", ngpu="4",
                                               shard_on="i"))
    assert cmds[0].split()[-4:] == ["--ngpu", "4", "--shard-on", "i"]
    cmds = eb.plugin_commands("contract", dict(spec="s", extents="e", input_a="a", input_b="b", output_c="c", shard_on="e"))
    assert cmds[0].split()[-2:] == ["--shard-on", "e"] and "--ngpu" not in cmds[0]  # one rank, still the sharded path


def test_cli_command_composition():
    line = eb.cli_command("/oge", "plugin", ["contract", "--spec", "adc,db=abc"], test=True)
    assert line == "execute_engine /oge --test --debug  --verbose  plugin  contract --spec adc,db=abc"
    assert eb.cli_command("/oge", "workflow", ["w"]).split() == ["execute_engine", "/oge", "--debug", "--verbose", "workflow", "w"]


def _engine_dry_run(plugin, params):
    argv = [ENGINE, eb.ENGINE_ROOT, "--test", "plugin", plugin]
    for k, v in params.items():
        argv += ["--" + k, v]
    log = subprocess.run(argv, capture_output=True, text=True, cwd=str(ROOT)).stdout
    return re.findall(r'Executing: "(.*)"', log)


@pytest.mark.skipif(not os.path.exists(ENGINE), reason="reference engine not built (oracle/build_ref.sh)")
def test_reference_engine_dry_run_equals_mirror():
    params = dict(spec="aeim,ebmj=abij", extents="a=64,b=64,e=64,i=64,j=64,m=64", input_a="/d/A", input_b="/d/B",
                  output_c="/d/C", alpha="2.5", shard="e:3/8")
    assert _engine_dry_run("contract", params) == eb.plugin_commands("contract", params)
    params = dict(spec="adij,bcdk=abcijk", extents="a=40,b=40,c=40,d=64,i=40,j=40,k=40", input_a="/d/A", input_b="/d/B",
                  output_c="/d/C", ngpu="8", shard_on="a")
    assert _engine_dry_run("contract", params) == eb.plugin_commands("contract", params)
    params = dict(spec="adc,db=abc", extents="a=8,b=6,c=5,d=7", output_plan="/d/plan.txt")
    assert _engine_dry_run("contract_plan", params) == eb.plugin_commands("contract_plan", params)


@pytest.mark.skipif(not os.path.exists(ENGINE), reason="reference engine not built (oracle/build_ref.sh)")
def test_plan_plugin_runs_on_cpu_through_the_engine(tmp_path, oracle):
    """BASELINE config 1 plumbing without a GPU: the engine system()s ctc_contract --describe."""
    out = tmp_path / "plan.txt"
    subprocess.run([ENGINE, eb.ENGINE_ROOT, "plugin", "contract_plan", "--spec", "adc,db=abc", "--extents",
                    "a=8,b=6,c=5,d=7", "--output_plan", str(out)], check=True, capture_output=True, cwd=str(tmp_path))
    text = out.read_text()
    o = oracle.classify("adc,db->abc")
    assert "canonical batch - m %s n %s k %s" % (o["m"], o["n"], o["k"]) in text
    assert "perm_a " + ",".join(map(str, o["perm_a"])) in text


@pytest.mark.gpu
@pytest.mark.parametrize("ext", [dict(a=8, b=6, c=5, d=7), dict(a=32, b=48, c=16, d=24)])
def test_config1_through_the_unchanged_engine(tmp_path, oracle, ext):
    """C[a,b,c] = A[a,d,c] * B[d,b]: files in, engine -> system(ctc_contract) on the GPU, file out."""
    if not os.path.exists(ENGINE):
        pytest.fail("oracle/_ref/execute_engine missing: run oracle/build_ref.sh where /root/reference exists")
    A = oracle.fill(ext["a"] * ext["d"] * ext["c"], 0xC7C0A)
    B = oracle.fill(ext["d"] * ext["b"], 0xC7C0B)
    A.tofile(tmp_path / "A.bin")
    B.tofile(tmp_path / "B.bin")
    extents = ",".join("%s=%d" % kv for kv in ext.items())
    run = subprocess.run([ENGINE, eb.ENGINE_ROOT, "plugin", "contract", "--spec", "adc,db=abc", "--extents", extents,
                          "--input_a", str(tmp_path / "A.bin"), "--input_b", str(tmp_path / "B.bin"),
                          "--output_c", str(tmp_path / "C.bin")], capture_output=True, text=True, cwd=str(tmp_path))
    assert os.path.exists(tmp_path / "C.bin"), run.stdout + run.stderr
    got = np.fromfile(tmp_path / "C.bin", dtype=np.float64)
    want = oracle.contract("adc,db->abc", ext, A, B)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,code,tol", [("f32", 1, 1e-5), ("bf16", 2, 8e-3)])
def test_other_element_types_through_the_unchanged_engine(tmp_path, oracle, dtype, code, tol):
    """--dtype travels through the plugin's template; files hold the raw element type."""
    if not os.path.exists(ENGINE):
        pytest.fail("oracle/_ref/execute_engine missing: run oracle/build_ref.sh where /root/reference exists")
    ext = dict(a=32, b=48, c=16, d=24)
    A = oracle.fill(ext["a"] * ext["d"] * ext["c"], 0xC7C0A, code)
    B = oracle.fill(ext["d"] * ext["b"], 0xC7C0B, code)
    A.tofile(tmp_path / "A.bin")
    B.tofile(tmp_path / "B.bin")
    extents = ",".join("%s=%d" % kv for kv in ext.items())
    run = subprocess.run([ENGINE, eb.ENGINE_ROOT, "plugin", "contract", "--spec", "adc,db=abc", "--extents", extents,
                          "--dtype", dtype, "--input_a", str(tmp_path / "A.bin"), "--input_b", str(tmp_path / "B.bin"),
                          "--output_c", str(tmp_path / "C.bin")], capture_output=True, text=True, cwd=str(tmp_path))
    assert os.path.exists(tmp_path / "C.bin"), run.stdout + run.stderr
    want = oracle.contract("adc,db->abc", ext, A, B, dtype=code)
    got = np.fromfile(tmp_path / "C.bin", dtype=want.dtype)
    assert got.shape == want.shape
    if dtype == "bf16":
        got, want = oracle.bf16_to_f32(got), oracle.bf16_to_f32(want)
    got, want = got.astype(np.float64), want.astype(np.float64)
    assert np.abs(got - want).max() <= tol * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("spec,ext,ngpu,shard_on", [
    ("aeim,ebmj=abij", dict(a=8, b=6, e=5, i=7, j=4, m=3), "2", "auto"),   # a: slowest mode of C, contiguous slabs
    ("aeim,ebmj=abij", dict(a=8, b=6, e=5, i=7, j=4, m=3), "3", "auto"),   # a does not divide by 3: b is taken
    ("aeim,ebmj=abij", dict(a=5, b=6, e=5, i=7, j=4, m=3), "4", "j"),      # fastest mode of C: interleaved runs
    ("adij,bcdk=abcijk", dict(a=4, b=3, c=2, d=16, i=6, j=5, k=8), "4", "k"),
    # contracted modes: slabs of A and B, partial results summed (RCCL with one rank per GPU; ranks that
    # share the box's one GPU are summed on the host; one rank runs the real one-rank RCCL reduce-scatter)
    ("aeim,ebmj=abij", dict(a=8, b=6, e=6, i=7, j=4, m=3), "2", "e"),
    ("aeim,ebmj=abij", dict(a=8, b=6, e=6, i=7, j=4, m=4), "4", "m"),
    ("aeim,ebmj=abij", dict(a=8, b=6, e=6, i=7, j=4, m=3), "1", "e"),
    ("adij,bcdk=abcijk", dict(a=4, b=3, c=2, d=16, i=6, j=5, k=8), "1", "d"),
])
def test_whole_contraction_on_several_gpus_through_the_unchanged_engine(tmp_path, oracle, spec, ext, ngpu, shard_on):
    """--ngpu: one process, one host thread per slab (ranks share the devices the box has), every rank
    holding only its slabs; the file must equal the oracle's unsharded result."""
    if not os.path.exists(ENGINE):
        pytest.fail("oracle/_ref/execute_engine missing: run oracle/build_ref.sh where /root/reference exists")
    ins, outm = spec.split("=")
    la, lb = ins.split(",")
    n = lambda m: int(np.prod([ext[c] for c in m]))  # noqa: E731
    A = oracle.fill(n(la), 0xC7C0A)
    B = oracle.fill(n(lb), 0xC7C0B)
    A.tofile(tmp_path / "A.bin")
    B.tofile(tmp_path / "B.bin")
    extents = ",".join("%s=%d" % kv for kv in ext.items())
    run = subprocess.run([ENGINE, eb.ENGINE_ROOT, "plugin", "contract", "--spec", spec, "--extents", extents,
                          "--input_a", str(tmp_path / "A.bin"), "--input_b", str(tmp_path / "B.bin"),
                          "--output_c", str(tmp_path / "C.bin"), "--ngpu", ngpu, "--shard_on", shard_on],
                         capture_output=True, text=True, cwd=str(tmp_path))
    assert os.path.exists(tmp_path / "C.bin"), run.stdout + run.stderr
    got = np.fromfile(tmp_path / "C.bin", dtype=np.float64)
    want = oracle.contract(spec.replace("=", "->"), ext, A, B)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


@pytest.mark.gpu
def test_several_gpus_with_beta_and_bad_requests(tmp_path, oracle):
    ext = dict(a=6, d=7, c=5, b=4)
    A = oracle.fill(6 * 7 * 5, 1)
    B = oracle.fill(7 * 4, 2)
    C0 = oracle.fill(6 * 4 * 5, 3)
    A.tofile(tmp_path / "A.bin")
    B.tofile(tmp_path / "B.bin")
    C0.tofile(tmp_path / "C.bin")
    base = [TOOL, "--spec", "adc,db=abc", "--extents", "a=6,d=7,c=5,b=4", "--a", str(tmp_path / "A.bin"), "--b",
            str(tmp_path / "B.bin"), "--out", str(tmp_path / "C.bin")]
    run = subprocess.run(base + ["--alpha", "0.5", "--beta", "-2", "--ngpu", "2", "--shard-on", "b"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    want = oracle.contract("adc,db->abc", ext, A, B, C=C0.copy(), alpha=0.5, beta=-2.0)
    got = np.fromfile(tmp_path / "C.bin", dtype=np.float64)
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    for extra in (["--ngpu", "2", "--shard-on", "d", "--beta", "1"], ["--ngpu", "7"], ["--ngpu", "2", "--shard", "a:0/2"],
                  ["--ngpu", "2", "--shard-on", "d"]):
        # beta with a contracted mode; nothing divides by 7; both kinds of sharding; 7 does not divide by 2
        run = subprocess.run(base + extra, capture_output=True, text=True)
        assert run.returncode != 0 and "ctc_contract:" in run.stderr, extra


@pytest.mark.gpu
def test_a_rank_that_fails_during_set_up_does_not_leave_its_peers_waiting(tmp_path, oracle):
    """Contracted mode over several ranks: the ranks meet after their set-up and enter the collectives (or the
    host-side sum) only if every one of them got there.  With one rank failing the tool must end at once with
    that rank's message, not hang in a collective its peer can never complete."""
    ext = dict(a=40, d=48, c=24, b=32)
    oracle.fill(40 * 48 * 24, 1).tofile(tmp_path / "A.bin")
    oracle.fill(48 * 32, 2).tofile(tmp_path / "B.bin")
    argv = [TOOL, "--spec", "adc,db=abc", "--extents", ",".join("%s=%d" % kv for kv in ext.items()), "--a", str(tmp_path / "A.bin"),
            "--b", str(tmp_path / "B.bin"), "--out", str(tmp_path / "C.bin"), "--ngpu", "4", "--shard-on", "d"]
    for failing in ("0", "2"):
        run = subprocess.run(argv, capture_output=True, text=True, timeout=60, env=dict(os.environ, CTC_CONTRACT_FAIL_RANK=failing))
        assert run.returncode != 0, run.stdout
        assert "rank %s: set-up failure requested" % failing in run.stderr, run.stderr
    run = subprocess.run(argv, capture_output=True, text=True, timeout=60)
    assert run.returncode == 0, run.stderr


def _engine_workflow_dry_run(workflow, params, cwd):
    argv = [ENGINE, eb.ENGINE_ROOT, "--test", "workflow", workflow]
    for k, v in params.items():
        argv += ["--" + k, v]
    log = subprocess.run(argv, capture_output=True, text=True, cwd=str(cwd)).stdout
    workdir = re.findall(r"Changing working directory to (\S+)", log)[0]
    commands = re.findall(r"Executing: (.*)", log)
    moves = re.findall(r"Moving file from (\S+) to (\S+)", log)
    return workdir, commands, moves


@pytest.mark.skipif(not os.path.exists(ENGINE), reason="reference engine not built (oracle/build_ref.sh)")
def test_reference_engine_workflow_dry_run_equals_mirror(tmp_path):
    """Two-step workflow (permute, then contract): temp-file names, step parameters and the final
    move as the unchanged engine logs them (engine/src/ctc_engine.cpp:186-311)."""
    params = {"input_a": "/d/A.bin", "input_b": "/d/B.bin", "modes": "adc=acd", "spec": "acd,db=abc",
              "extents": "a=8,b=6,c=5,d=7", "output_c": "/d/C.bin", "perm.dtype": "f32", "mult.dtype": "f32",
              "mult.alpha": "0.5"}
    workdir, commands, moves = _engine_workflow_dry_run("permute_contract", params, tmp_path)
    want_cmds, want_moves = eb.workflow_commands("permute_contract", params, workdir)
    assert commands == want_cmds
    assert [tuple(m) for m in moves] == want_moves
    assert "--a perm-permute-output_t.tmp.tensor" in commands[1]  # later steps name the file without the directory
    # a relative final output lands in the caller's directory
    params["output_c"] = "C.bin"
    workdir, commands, moves = _engine_workflow_dry_run("permute_contract", params, tmp_path)
    assert moves == [("mult-contract-output_c.tmp.tensor", str(tmp_path) + "/C.bin")]


def test_workflow_mirror_rejects_unknown_step():
    with pytest.raises(eb.EngineError):
        eb.workflow_commands("permute_contract", {"nostep.dtype": "f32"}, "/W")


@pytest.mark.gpu
def test_permute_then_contract_workflow_through_the_unchanged_engine(tmp_path, oracle):
    """A stored as [a,d,c] is permuted to [a,c,d] by step 1 and contracted by step 2; the engine
    moves the result to the requested file."""
    if not os.path.exists(ENGINE):
        pytest.fail("oracle/_ref/execute_engine missing: run oracle/build_ref.sh where /root/reference exists")
    ext = dict(a=32, b=48, c=16, d=24)
    A = oracle.fill(ext["a"] * ext["d"] * ext["c"], 0xC7C0A)
    B = oracle.fill(ext["d"] * ext["b"], 0xC7C0B)
    A.tofile(tmp_path / "A.bin")
    B.tofile(tmp_path / "B.bin")
    extents = ",".join("%s=%d" % kv for kv in ext.items())
    run = subprocess.run([ENGINE, eb.ENGINE_ROOT, "workflow", "permute_contract", "--modes", "adc=acd", "--spec", "acd,db=abc",
                          "--extents", extents, "--input_a", str(tmp_path / "A.bin"), "--input_b", str(tmp_path / "B.bin"),
                          "--output_c", str(tmp_path / "C.bin")], capture_output=True, text=True, cwd=str(tmp_path))
    assert os.path.exists(tmp_path / "C.bin"), run.stdout + run.stderr
    got = np.fromfile(tmp_path / "C.bin", dtype=np.float64)
    want = oracle.contract("adc,db->abc", ext, A, B)   # same result as contracting the unpermuted A
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


# ---- what the reference itself holds: its fastr plugin, its bwa_pe_sam workflow, its C++ example ----------
REFERENCE_OGE = "/root/reference/engine/test"
ALIGN_EXAMPLE = os.path.join(ROOT, "oracle", "_ref", "align_workflow")
IMPORT_SCRIPT = os.path.join(ROOT, "oracle", "_ref", "import_script")


def _reference_golden():
    with open(os.path.join(ROOT, "tests", "golden", "boundary_reference_golden.json")) as f:
        return json.load(f)


def _align_example_workflow():
    """The workflow engine/examples/c++/align_workflow.cpp:23-62 builds through the C++ API."""
    wf = eb.Workflow()
    wf.inputs = {"in_read1": {}, "in_read2": {}, "in_genome": {}}
    wf.add_step("aln1", "bwa_aln", [("input_read", "$in_read1"), ("ref_genome", "$in_genome")], ["output_sai"])
    wf.add_step("aln2", "bwa_aln", [("input_read", "$in_read2"), ("ref_genome", "$in_genome")], ["output_sai"])
    wf.add_step("tosam", "bwa_sampe", [("input_read1", "$in_read1"), ("input_read2", "$in_read2"), ("ref_genome", "$in_genome"),
                                       ("input_sai1", "$aln1.output_sai"), ("input_sai2", "$aln2.output_sai")], ["output_sam"])
    wf.add_step("tobam", "sam_sorted_bam", [("input_sam", "$tosam.output_sam")], ["output_bam"])
    wf.outputs = {"output": "$tobam.output_bam"}
    params = {"in_read1": "/path/to/in1.fast1", "in_read2": "/path/to/in1.fast2", "in_genome": "/path/to/genome.fasta",
              "output": "/path/to/results.bam"}
    return wf, params


@pytest.mark.skipif(not os.path.isdir(REFERENCE_OGE), reason="needs the reference's plugin repository (/root/reference)")
def test_mirror_reproduces_the_reference_held_fixtures_byte_for_byte():
    """The command lines the reference engine logged for ITS OWN descriptors (tests/golden/
    boundary_reference_golden.json, made by oracle/make_boundary_golden.py) against the host-side mirror
    reading the same descriptors: equal but for the mkdtemp directory, which reads <WORK>."""
    for c in _reference_golden():
        if c["kind"] == "plugin":
            got = ['"%s"' % x for x in eb.plugin_commands(c["id"], c["params"], engine_root=REFERENCE_OGE)]
            assert got == c["commands"], c["id"]
            continue
        if c["kind"] == "workflow":
            cmds, moves = eb.workflow_commands(c["id"], c["params"], "<WORK>", engine_root=REFERENCE_OGE)
        else:
            wf, params = _align_example_workflow()
            cmds, moves = eb.workflow_commands_of(wf, params, "<WORK>", engine_root=REFERENCE_OGE)
        assert cmds == c["commands"], c["id"]
        assert [list(m) for m in moves] == c["moves"], c["id"]


@pytest.mark.skipif(not (os.path.isdir(REFERENCE_OGE) and os.path.exists(ALIGN_EXAMPLE) and os.path.exists(ENGINE)),
                    reason="needs /root/reference and oracle/_ref (oracle/build_ref.sh)")
def test_reference_engine_still_logs_the_committed_fixtures(tmp_path):
    """Live: the reference engine and the reference's example, compiled here unmodified, on a scratch copy
    of the reference's repository, against the committed fixture."""
    import shutil
    root = str(tmp_path / "oge")
    shutil.copytree(REFERENCE_OGE, root)
    for c in _reference_golden():
        if c["kind"] == "example":
            argv = [ALIGN_EXAMPLE, root]
        else:
            argv = [ENGINE, root, "--test", c["kind"], c["id"]]
            for k, v in c["params"].items():
                argv += ["--" + k, v]
        log = subprocess.run(argv, capture_output=True, text=True, cwd=root).stdout
        work = re.findall(r"Changing working directory to (\S+)", log)[0]
        cmds = [x.replace(work, "<WORK>").replace(root, "<ROOT>") for x in re.findall(r"Executing: (.*)", log)]
        moves = [[a.replace(work, "<WORK>"), b] for a, b in re.findall(r"Moving file from (\S+) to (\S+)", log)]
        assert cmds == c["commands"] and moves == c["moves"], c["id"]


def _imported_script(tool, b_file, ext):
    extents = ",".join("%s=%d" % kv for kv in ext.items())
    return "\n".join([
        "# two inline plugins chained by the reference's script importer",
        "WORKFLOW_ID=imported_permute_contract",
        'WORKFLOW_NAME="permute A, then contract, from a script"',
        "WORKFLOW_VERSION=0.1",
        "SUITE_ID=imported",
        "TOOL=" + tool,
        "B_FILE=" + b_file,
        "permute_a := {",
        "    $TOOL --permute adc=acd --extents " + extents + " --dtype f64 --a $input --out $output",
        "}",
        "contract_ab := {",
        "    $TOOL --spec acd,db=abc --extents " + extents + " --dtype f64 --a $input --b $B_FILE --out $output",
        "}",
        ""])


@pytest.mark.skipif(not (os.path.exists(IMPORT_SCRIPT) and os.path.exists(ENGINE)), reason="needs oracle/_ref (oracle/build_ref.sh)")
def test_import_script_builds_the_permute_contract_workflow(tmp_path):
    """The reference's import_script (engine/src/import_script.cpp:50-97), compiled unmodified, turns a
    script with two inline definitions into two plugins and a workflow inside a scratch copy of the
    contraction repository; the unchanged engine then dry-runs it: each step's $input is the previous
    step's $output, the first $input and the last $output are the workflow's."""
    import shutil
    root = str(tmp_path / "oge")
    shutil.copytree(eb.ENGINE_ROOT, root, ignore=shutil.ignore_patterns("workspace"))
    ext = dict(a=8, b=6, c=5, d=7)
    script = tmp_path / "script.txt"
    script.write_text(_imported_script(TOOL, "/d/B.bin", ext))
    run = subprocess.run([IMPORT_SCRIPT, root, str(script)], capture_output=True, text=True, cwd=str(tmp_path))
    assert os.path.exists(os.path.join(root, "workflows", "imported_permute_contract.xml")), run.stdout + run.stderr
    log = subprocess.run([ENGINE, root, "--test", "workflow", "imported_permute_contract", "--input", "/d/A.bin", "--output",
                          "/d/C.bin"], capture_output=True, text=True, cwd=str(tmp_path)).stdout
    work = re.findall(r"Changing working directory to (\S+)", log)[0]
    cmds = re.findall(r"Executing: (.*)", log)
    extents = "a=8,b=6,c=5,d=7"
    assert [c.strip() for c in cmds] == [
        "%s --permute adc=acd --extents %s --dtype f64 --a /d/A.bin --out %s/permute_a-permute_a-output.tmp" % (TOOL, extents, work),
        "%s --spec acd,db=abc --extents %s --dtype f64 --a permute_a-permute_a-output.tmp --b /d/B.bin --out "
        "%s/contract_ab-contract_ab-output.tmp" % (TOOL, extents, work)]
    assert re.findall(r"Moving file from (\S+) to (\S+)", log) == [("contract_ab-contract_ab-output.tmp", "/d/C.bin")]


@pytest.mark.gpu
def test_imported_script_workflow_runs_on_the_gpu_through_the_unchanged_engine(tmp_path, oracle):
    """The same imported workflow executed for real: A stored as [a,d,c] is permuted and contracted by the
    two script-defined plugins; the file the engine moves into place must equal the oracle's result."""
    import shutil
    if not (os.path.exists(IMPORT_SCRIPT) and os.path.exists(ENGINE)):
        pytest.fail("oracle/_ref/import_script or execute_engine missing: run oracle/build_ref.sh where /root/reference exists")
    root = str(tmp_path / "oge")
    shutil.copytree(eb.ENGINE_ROOT, root, ignore=shutil.ignore_patterns("workspace"))
    ext = dict(a=32, b=48, c=16, d=24)
    A = oracle.fill(ext["a"] * ext["d"] * ext["c"], 0xC7C0A)
    B = oracle.fill(ext["d"] * ext["b"], 0xC7C0B)
    A.tofile(tmp_path / "A.bin")
    B.tofile(tmp_path / "B.bin")
    script = tmp_path / "script.txt"
    script.write_text(_imported_script(TOOL, str(tmp_path / "B.bin"), ext))
    subprocess.run([IMPORT_SCRIPT, root, str(script)], capture_output=True, text=True, cwd=str(tmp_path))
    run = subprocess.run([ENGINE, root, "workflow", "imported_permute_contract", "--input", str(tmp_path / "A.bin"), "--output",
                          str(tmp_path / "C.bin")], capture_output=True, text=True, cwd=str(tmp_path))
    assert os.path.exists(tmp_path / "C.bin"), run.stdout + run.stderr
    got = np.fromfile(tmp_path / "C.bin", dtype=np.float64)
    want = oracle.contract("adc,db->abc", ext, A, B)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
