"""Minimal stand-ins for ml_collections.ConfigDict / config_flags and absl flags.

The reference reads `--config=<file.py>` exposing get_config() plus dotted `--config.a.b=v` overrides
(ldm/main.py:29-36, ldm/eval_bpd.py:17-31).  ml_collections and absl are not installed on the target
image, so this module provides the subset the path needs; a reference-style config file that does
`import ml_collections` keeps working because `load_config_file` installs a shim module of that name
when the real one is absent.
"""
import ast
import importlib.util
import os
import sys
import types


class ConfigDict:
    def __init__(self, initial_dictionary=None, **kwargs):
        object.__setattr__(self, "_fields", {})
        init = dict(initial_dictionary or {})
        init.update(kwargs)
        for k, v in init.items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        return ConfigDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return object.__getattribute__(self, "_fields")[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self._fields[k] = self._wrap(v)

    __setitem__ = __setattr__

    def __getitem__(self, k):
        return self._fields[k]

    def __contains__(self, k):
        return k in self._fields

    def __iter__(self):
        return iter(self._fields)

    def keys(self):
        return self._fields.keys()

    def items(self):
        return self._fields.items()

    def values(self):
        return self._fields.values()

    def get(self, k, default=None):
        return self._fields.get(k, default)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self._fields.items()}

    def __repr__(self):
        return f"ConfigDict({self.to_dict()!r})"

    def __deepcopy__(self, memo):
        import copy
        return ConfigDict(copy.deepcopy(self.to_dict(), memo))

    def set_path(self, dotted, raw):
        """--config.a.b=raw : the value is parsed with the type of the existing field when there is one."""
        keys = dotted.split(".")
        node = self
        for k in keys[:-1]:
            if k not in node:
                node[k] = ConfigDict()
            node = node[k]
        old = node.get(keys[-1], None)
        node[keys[-1]] = _parse_value(raw, old)


def _parse_value(raw, old):
    if isinstance(old, bool):
        if raw.lower() in ("true", "1", "yes"):
            return True
        if raw.lower() in ("false", "0", "no"):
            return False
        raise ValueError(f"not a bool: {raw}")
    if isinstance(old, int) and not isinstance(old, bool):
        try:
            return int(raw.replace("_", ""))
        except ValueError:
            return float(raw)
    if isinstance(old, float):
        return float(raw)
    if isinstance(old, str):
        return raw
    try:
        return ast.literal_eval(raw)
    except (ValueError, SyntaxError):
        return raw


def _install_ml_collections_shim():
    try:
        import ml_collections  # noqa: F401
        return
    except ImportError:
        pass
    shim = types.ModuleType("ml_collections")
    shim.ConfigDict = ConfigDict
    shim.__doc__ = "shim installed by mulan_amd.config (ml_collections is not installed)"
    sys.modules["ml_collections"] = shim


def load_config_file(path):
    _install_ml_collections_shim()
    path = os.path.abspath(path)
    spec = importlib.util.spec_from_file_location("_mulan_config_" + str(abs(hash(path))), path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = mod.get_config()
    if isinstance(cfg, dict):
        cfg = ConfigDict(cfg)
    return cfg


class Flags:
    """Tiny absl.flags look-alike: DEFINE_* then parse(argv) -> attribute access."""

    def __init__(self):
        self._defs = {}
        self._vals = {}
        self._required = []

    def DEFINE(self, name, default, typ, help_=""):
        self._defs[name] = (typ, help_)
        self._vals[name] = default

    def DEFINE_string(self, name, default, help_=""):
        self.DEFINE(name, default, str, help_)

    def DEFINE_integer(self, name, default, help_=""):
        self.DEFINE(name, default, int, help_)

    def DEFINE_float(self, name, default, help_=""):
        self.DEFINE(name, default, float, help_)

    def DEFINE_bool(self, name, default, help_=""):
        self.DEFINE(name, default, bool, help_)

    def DEFINE_config_file(self, name, default=None, help_=""):
        self.DEFINE(name, default, "config", help_)

    def mark_flags_as_required(self, names):
        self._required += list(names)

    def parse(self, argv):
        overrides = []
        i = 0
        args = list(argv)
        while i < len(args):
            a = args[i]
            i += 1
            if not a.startswith("--"):
                raise SystemExit(f"unexpected argument {a!r}")
            body = a[2:]
            if "=" in body:
                key, val = body.split("=", 1)
            elif body.startswith("no") and body[2:] in self._defs and self._defs[body[2:]][0] is bool:
                key, val = body[2:], "false"
            elif body in self._defs and self._defs[body][0] is bool:
                key, val = body, "true"
            else:
                if i >= len(args):
                    raise SystemExit(f"flag --{body} needs a value")
                key, val = body, args[i]
                i += 1
            head = key.split(".", 1)[0]
            if head not in self._defs:
                raise SystemExit(f"unknown flag --{head}")
            typ = self._defs[head][0]
            if "." in key:
                if typ != "config":
                    raise SystemExit(f"--{key}: only config flags take dotted overrides")
                overrides.append((key.split(".", 1)[1], val, head))
            elif typ == "config":
                self._vals[head] = load_config_file(val)
            elif typ is bool:
                self._vals[key] = _parse_value(val, False)
            else:
                self._vals[key] = typ(val)
        for dotted, val, head in overrides:
            if self._vals[head] is None:
                raise SystemExit(f"--{head}.{dotted} given before --{head}=<file>")
            self._vals[head].set_path(dotted, val)
        for r in self._required:
            if self._vals.get(r) is None:
                raise SystemExit(f"flag --{r} is required")
        return self

    def __getattr__(self, k):
        vals = object.__getattribute__(self, "_vals")
        if k in vals:
            return vals[k]
        raise AttributeError(k)
