"""Host-side mirror of the CTC engine's plugin boundary, for the contraction plugins.

The engine itself stays unchanged (it is the host); this module restates, for tests
and for callers that want the command line without spawning the engine, exactly what
the engine does between `executePlugin` and `system()`:

  * plugin XML -> inputs/outputs/commands        engine/src/ctc_plugin.cpp:16-109
  * parameter check + defaults                   engine/src/ctc_plugin.cpp:175-198
  * template expansion ($v, ${v}, "...", [c ? a : b])   engine/src/ctc_util.cpp:130-290
  * the command line the interactive CLI composes       engine/cli/cli.py:98-133

Errors are raised as EngineError where the engine throws a C string.
"""
import os
import xml.etree.ElementTree as ET

_WHITE = " \t\n\r"

ENGINE_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "engine")


class EngineError(RuntimeError):
    pass


def resolve_word(word, exec_conf, sys_conf, env=None):
    """One whitespace-free token: quoted text passes through; $name / ${name} are looked up in
    the call parameters, then the engine's system parameters, then the environment."""
    env = os.environ if env is None else env
    word = "".join(ch for ch in word if not ch.isspace())
    if word.startswith('"'):
        close = word.find('"', 1)
        if close < 0:
            raise EngineError("Error resolving words")
        return word[1:close]
    out, pos = [], 0
    while True:
        dollar = word.find("$", pos)
        if dollar < 0:
            out.append(word[pos:])
            break
        out.append(word[pos:dollar])
        if word[dollar + 1:dollar + 2] == "{":
            close = word.find("}", dollar + 1)
            if close < 0:
                raise EngineError("Error resolving words")
            name, pos = word[dollar + 2:close], close + 1
        else:
            end = dollar + 1
            while end < len(word) and word[end] not in "/\\$":
                end += 1
            name, pos = word[dollar + 1:end], end
        if name in exec_conf:
            out.append(exec_conf[name])
        elif name in sys_conf:
            out.append(sys_conf[name])
        elif name in env:
            out.append(env[name])
        else:
            raise EngineError("Error resolving variables: " + name)
    return "".join(out)


def resolve_conditional(text, exec_conf, sys_conf, env=None):
    """[cond ? a : b], [cond ? a]; cond is `x == y`, `x != y`, or a value that must be "true"."""
    q = text.find("?")
    if q < 0:
        raise EngineError("Error parsing conditional")
    cond, value = text[:q], text[q + 1:]
    at = cond.find("==")
    if at < 0:
        at = cond.find("!=")
    if at >= 0:
        op = cond[at:at + 2]
        lhs = resolve_word(cond[:at], exec_conf, sys_conf, env)
        rhs = resolve_word(cond[at + 2:], exec_conf, sys_conf, env)
        truth = (lhs == rhs) if op == "==" else (lhs != rhs)
    else:
        truth = resolve_word(cond, exec_conf, sys_conf, env) == "true"
    colon = value.find(":")
    if colon >= 0:
        if truth:
            # the engine drops the character before ':' (templates put a space there)
            return resolve_stream(value[:max(colon - 1, 0)], exec_conf, sys_conf, env)
        return resolve_word(value[colon + 1:], exec_conf, sys_conf, env)
    return resolve_stream(value, exec_conf, sys_conf, env) if truth else ""


def resolve_stream(text, exec_conf, sys_conf, env=None):
    """A whole command template: tokens separated by single spaces in the result."""
    out, pos, n = [], 0, len(text)
    while True:
        if pos < n and text[pos] in _WHITE:
            out.append(" ")
        start = pos
        while start < n and text[start] in _WHITE:
            start += 1
        if start >= n:
            break
        if text[start] == "[":
            close = text.find("]", start + 1)
            if close < 0:
                raise EngineError("error parsing command")
            out.append(resolve_conditional(text[start + 1:close], exec_conf, sys_conf, env))
            pos = close + 1
        elif text[start] == '"':
            close = text.find('"', start + 1)
            if close < 0:
                raise EngineError("error parsing command")
            out.append(text[start:close + 1])
            pos = close + 1
        else:
            end = start
            while end < n and text[end] not in _WHITE:
                end += 1
            out.append(resolve_word(text[start:end], exec_conf, sys_conf, env))
            pos = end
    return "".join(out)


class Plugin:
    """A plugin descriptor as the engine reads it."""

    def __init__(self, path):
        self.path = os.path.abspath(path)
        self.conf_dir = os.path.dirname(self.path) + "/"
        root = ET.parse(self.path).getroot()
        self.id = root.attrib["id"]
        self.name = root.attrib["name"]
        self.version = root.attrib["version"]
        self.description = root.findtext("description")
        self.commands = [c.text or "" for c in root.find("commands")]
        self.inputs, self.outputs = {}, {}
        for p in root.find("inputs"):
            if p.attrib["type"] == "file" and "format" not in p.attrib:
                raise EngineError("file parameter without format: " + p.attrib["name"])
            self.inputs[p.attrib["name"]] = {
                "type": p.attrib["type"], "format": p.attrib.get("format", ""),
                "value": p.attrib.get("value", ""), "label": p.attrib["label"]}
        for p in root.find("outputs"):
            self.outputs[p.attrib["name"]] = {
                "type": p.attrib.get("type", "file"), "format": p.attrib.get("format", ""),
                "value": "", "label": p.attrib["label"]}

    def exec_commands(self, params, sys_params=None, env=None):
        """The shell commands the engine would run for these call parameters."""
        sys_params = dict(sys_params or {})
        sys_params["PLUGIN_DIR"] = self.conf_dir
        params = dict(params)
        for name in params:
            if name not in self.inputs and name not in self.outputs:
                raise EngineError("Plugin::getExecCommand(): Invalid parameters! " + name)
        for name, p in list(self.inputs.items()) + list(self.outputs.items()):
            params.setdefault(name, p["value"])
        return [resolve_stream(c, params, sys_params, env) for c in self.commands]


def catalog_plugin_file(engine_root, plugin_id):
    """File of a plugin id, from <engine_root>/catalog.xml (engine/src/ctc_catalog.cpp:28-72)."""
    root = ET.parse(os.path.join(engine_root, "catalog.xml")).getroot()
    for suite in root:
        for item in suite:
            if item.tag == "plugin" and item.attrib.get("id") == plugin_id:
                return os.path.join(engine_root, item.attrib["file"])
    raise EngineError("plugin not in catalog: " + plugin_id)


def plugin_commands(plugin_id, params, engine_root=ENGINE_ROOT, env=None):
    root = os.path.abspath(engine_root)
    if not root.endswith("/"):
        root += "/"
    plugin = Plugin(catalog_plugin_file(root, plugin_id))
    return plugin.exec_commands(params, {"ENGINE_DIR": root}, env)


def _catalog_file(engine_root, tag, item_id):
    root = ET.parse(os.path.join(engine_root, "catalog.xml")).getroot()
    for suite in root:
        for item in suite:
            if item.tag == tag and item.attrib.get("id") == item_id:
                return os.path.join(engine_root, item.attrib["file"])
    raise EngineError("%s not in catalog: %s" % (tag, item_id))


class Workflow:
    """A workflow descriptor as the engine reads it (engine/src/ctc_workflow.cpp:38-144), or one put
    together step by step as a C++ host does with Workflow::addInput / addStep / addOutput
    (engine/examples/c++/align_workflow.cpp:23-62)."""

    def __init__(self, path=None):
        if path is None:
            self.id, self.inputs, self.steps, self.outputs = "", {}, [], {}
            return
        root = ET.parse(path).getroot()
        self.id = root.attrib["id"]
        self.inputs = {p.attrib["name"]: dict(p.attrib) for p in root.find("inputs")}
        self.steps = []
        for st in root.find("steps"):
            if st.attrib["type"] != "plugin":
                raise EngineError("only plugin steps are mirrored here")
            self.steps.append({
                "id": st.attrib["id"], "plugin_id": st.attrib["plugin_id"],
                "inputs": [(p.attrib["name"], p.attrib["value"]) for p in st.find("inputs")],
                "outputs": [p.attrib["name"] for p in st.find("outputs")]})
        self.outputs = {}
        for p in root.find("outputs"):
            if "type" not in p.attrib:
                raise EngineError("Missing \"type\" in the workflow definition")
            self.outputs[p.attrib["name"]] = p.attrib["value"]


    def add_step(self, step_id, plugin_id, inputs, outputs):
        """inputs: (name, value) pairs; outputs: names."""
        self.steps.append({"id": step_id, "plugin_id": plugin_id, "inputs": list(inputs), "outputs": list(outputs)})
        return self


def workflow_commands(workflow_id, params, workdir, engine_root=ENGINE_ROOT, org_dir="/", env=None):
    """workflow_commands_of for a workflow of the catalog."""
    root = os.path.abspath(engine_root)
    wf = Workflow(_catalog_file(root + "/", "workflow", workflow_id))
    return workflow_commands_of(wf, params, workdir, engine_root, org_dir, env)


def workflow_commands_of(wf, params, workdir, engine_root=ENGINE_ROOT, org_dir="/", env=None):
    """What Engine::executeWorkflow does with these call parameters
    (engine/src/ctc_engine.cpp:186-311): returns (commands, moves).  `workdir` stands for the
    mkdtemp directory the engine changes into; `name.param` call parameters go to step `name`;
    each step output gets <workdir>/<step>-<plugin>-<param>.tmp[.format], and later steps see
    it as $<step>.<param> WITHOUT the directory (they run inside it)."""
    root = os.path.abspath(engine_root)
    if not root.endswith("/"):
        root += "/"
    sys_params = {"ENGINE_DIR": root}
    exec_params, step_params = {}, {st["id"]: {} for st in wf.steps}
    for name, value in params.items():
        if "." in name:
            sid, rest = name.split(".", 1)
            if sid not in step_params:
                raise EngineError("Cannot find matched step for parameter " + name)
            step_params[sid][rest] = value
        else:
            exec_params[name] = value
    commands = []
    for st in wf.steps:
        sp = step_params[st["id"]]
        # std::map iteration order: by parameter name
        for name, value in sorted(st["inputs"]):
            if name not in sp:
                sp[name] = resolve_word(value, exec_params, sys_params, env)
        plugin = Plugin(catalog_plugin_file(root, st["plugin_id"]))
        for name in sorted(st["outputs"]):
            fmt = plugin.outputs.get(name, plugin.inputs.get(name, {"format": ""}))["format"]
            tmp = "%s-%s-%s.tmp" % (st["id"], st["plugin_id"], name) + ("." + fmt if fmt else "")
            sp[name] = workdir + "/" + tmp
            exec_params["%s.%s" % (st["id"], name)] = tmp
        commands += plugin.exec_commands(sp, sys_params, env)
    moves = []
    for name in sorted(wf.outputs):
        source = resolve_word(wf.outputs[name], exec_params, sys_params, env)
        dest = exec_params.get(name, "")
        if not dest.startswith("/"):
            dest = org_dir + "/" + dest
        moves.append((source, dest))
    return commands, moves


def cli_command(engine_root, kind, words, test=False):
    """The line the engine's interactive CLI hands to os.system for executePlugin/testPlugin/
    executeWorkflow/testWorkflow (engine/cli/cli.py:98-133)."""
    if kind not in ("plugin", "workflow"):
        raise ValueError(kind)
    cmd = "execute_engine " + engine_root
    if test:
        cmd += " --test"
    cmd += " --debug "
    cmd += " --verbose "
    cmd += " %s " % kind
    for w in words:
        cmd += " " + w
    return cmd
