"""Text scene -> record buffers. Mirrors the reference's `SceneLoader::Load(path, objects&, lights&)`
(SceneLoader.hpp:14, SceneLoader.cpp:32-331; grammar in SURVEY.md Appendix A):

    header:  `material <name>` / `light <name>` blocks at indent 0, properties at indent 2
             material: ambient|diffuse|specular r g b, absorption|reflection|transparency|shininess v
             light:    ambient|diffuse|specular r g b
    `===`
    body:    `primative <sphere|box> <material>` (sic), `light <name>`,
             `translate x y z` | `scale x y z` | `rotate deg ax ay az` - each pushes top*T on a matrix
             stack and scopes the following lines indented two more spaces.

The stack is seeded with lookAt(eye (0,0,10), centre 0, up +y) (SceneLoader.cpp:211-216), so all
emitted coordinates are view space. Errors raise SceneError with the reference's 1-based line
numbers and message wording; an unknown material/light name raises KeyError (std::out_of_range
from map::at in the reference, SceneLoader.cpp:254,261).
"""
from __future__ import annotations

import numpy as np

from . import records as R


class SceneError(RuntimeError):
    """std::runtime_error in the reference."""


class SceneLoader:
    def __init__(self):
        self.lines: list[str] = []
        self.materials: dict[str, R.Material] = {}
        self.light_properties: dict[str, R.LightProperties] = {}
        self.line_num = 0
        self.last_indent = 0

    # -- public ------------------------------------------------------------------------
    def Load(self, scene_file: str):
        """Returns (ObjectData records, Light records) as numpy arrays in emission order."""
        try:
            with open(scene_file) as f:
                text = f.read()
        except OSError:
            raise SceneError(f"Scene file '{scene_file}' could not be found.")
        return self.LoadString(text)

    def LoadString(self, text: str):
        self.lines = text.split("\n")
        if self.lines and self.lines[-1] == "":
            self.lines.pop()  # getline() does not produce a trailing empty line
        self.lines = [ln.rstrip("\r") for ln in self.lines]
        self._parse_header()
        objs, lights = self._parse_body()
        return R.objects_array(objs), R.lights_array(lights)

    # -- helpers -----------------------------------------------------------------------
    def _err(self, msg: str):
        return SceneError(f"Error parsing scene file at line {self.line_num}:\n\t{msg}")

    def _next_line(self):
        """GetNextLine (SceneLoader.cpp:305-331): skip blank/comment lines, validate indentation."""
        while self.line_num < len(self.lines):
            line = self.lines[self.line_num]
            self.line_num += 1
            if not line:
                continue
            stripped = line.lstrip(" ")
            if not stripped:
                continue
            first = len(line) - len(stripped)
            if stripped[0] == "#":
                continue
            if first % 2 != 0:
                raise self._err("line does not have proper indentation, must be multiples of two")
            if first > self.last_indent:
                raise self._err("line is indented too far")
            return line, first
        return None

    def _floats(self, toks, n, name, usage):
        vals = []
        for i in range(n):
            try:
                vals.append(float(np.float32(toks[1 + i])))
            except (IndexError, ValueError):
                found = i + 1 if n > 1 else 0
                plural = "s" if n > 1 else ""
                raise self._err(f"{name} expects {n} argument{plural}, found {found}\n\t{usage}")
        return vals

    def _parse_header(self):
        state = None  # None | ("material", name) | ("light", name)
        while True:
            got = self._next_line()
            if got is None:
                return
            line, indent = got
            if line == "===":
                return
            while self.last_indent > indent:
                self.last_indent -= 2
                state = None
            toks = line.split()
            cmd = toks[0]
            if state is None:
                if cmd in ("material", "light"):
                    self.last_indent += 2
                    if len(toks) < 2:
                        raise self._err(f"{cmd} expects 1 argument, found 0\n\t{cmd} <{cmd} name>")
                    name = toks[1]
                    if cmd == "material":
                        self.materials.setdefault(name, R.Material())
                    else:
                        self.light_properties.setdefault(name, R.LightProperties())
                    state = (cmd, name)
                else:
                    raise self._err(f"unsupported command '{cmd}' in header\n\tif you are trying to specify "
                                    "properties, ensure the correct level of indentation")
                continue
            kind, name = state
            target = self.materials[name] if kind == "material" else self.light_properties[name]
            if cmd in ("ambient", "diffuse", "specular"):
                setattr(target, cmd, tuple(self._floats(toks, 3, cmd, f"{cmd} <r> <g> <b>")))
            elif kind == "material" and cmd in ("absorption", "reflection", "transparency", "shininess"):
                what = "shininess value" if cmd == "shininess" else f"{cmd} ratio"
                setattr(target, cmd, self._floats(toks, 1, cmd, f"{cmd} <{what}>")[0])
            elif cmd in ("material", "light"):
                raise self._err(f"tried to declare a {cmd} in a nested scope, unindent to declare a new {cmd}")
            else:
                raise self._err(f"unsupported command '{cmd}' while parsing {kind}")

    def _parse_body(self):
        objs, lights = [], []
        root = R.mat_mul(R.mat_identity(), R.look_at((0, 0, 10), (0, 0, 0), (0, 1, 0)))
        stack = [root, root.copy()]
        while True:
            got = self._next_line()
            if got is None:
                break
            line, indent = got
            while self.last_indent > indent:
                self.last_indent -= 2
                stack.pop()
            toks = line.split()
            cmd = toks[0]
            if cmd == "primative":
                if len(toks) < 2:
                    raise self._err("primative expects 2 argument, found 0\n\tprimative <primative type> <material name>")
                if len(toks) < 3:
                    raise self._err("primative expects 2 argument, found 1\n\tprimative <primative type> <material name>")
                if toks[1] == "sphere":
                    ptype = R.SPHERE
                elif toks[1] == "box":
                    ptype = R.BOX
                else:
                    raise self._err(f"unsupported primative type '{toks[1]}'")
                objs.append(R.make_object(ptype, self.materials[toks[2]], stack[-1]))
            elif cmd == "light":
                if len(toks) < 2:
                    raise self._err("light expects 1 argument, found 0\n\tlight <light name>")
                lights.append(R.make_light(self.light_properties[toks[1]], stack[-1]))
            elif cmd == "translate":
                v = self._floats(toks, 3, "translate", "translate <x> <y> <z>")
                stack.append(R.mat_mul(stack[-1], R.translate(R.mat_identity(), v)))
                self.last_indent += 2
            elif cmd == "scale":
                v = self._floats(toks, 3, "scale", "scale <x> <y> <z>")
                stack.append(R.mat_mul(stack[-1], R.scale(R.mat_identity(), v)))
                self.last_indent += 2
            elif cmd == "rotate":
                v = self._floats(toks, 4, "rotate", "rotate <angle in degrees> <axis x> <axis y> <axis z>")
                rot = R.rotate(R.mat_identity(), R.radians(v[0]), R.normalize3(v[1:4]))
                stack.append(R.mat_mul(stack[-1], rot))
                self.last_indent += 2
            else:
                raise self._err(f"unsupported command '{cmd}' in body")
        return objs, lights


def load_scene(path: str):
    return SceneLoader().Load(path)
