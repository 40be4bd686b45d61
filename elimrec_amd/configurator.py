"""`.properties` + `--key=value` configuration surface of the reference (util/configurator.py:11-158).

Resolution rules kept identical because drivers depend on them:
  * library file (NeuRec.properties) -> `lib_arg`; `conf/<recommender>.properties` -> `alg_arg`;
    `--name=value` argv tokens -> `cmd_arg` (each must start with `--` and hold exactly one `=`);
  * a command-line value overrides a file value only for keys already present in that file
    (:98-100); lookup order is lib, alg, cmd (:121-128) so CLI-only keys (alpha, loss,
    predict_type, modality, ...) still resolve;
  * values are python-evaluated to int/float/list/tuple/bool/None, `true`/`false` in any case
    become booleans, everything else stays a string (:131-143);
  * attribute access aliases item access; a missing key raises KeyError.
"""
import os
import sys
from collections import OrderedDict
from configparser import ConfigParser

_SCALARS = (str, int, float, list, tuple, bool, type(None))


def _parse_value(text):
    try:
        value = eval(text)  # noqa: S307 -- same permissive parser as the reference (numbers, lists, None, 'quoted')
        return value if isinstance(value, _SCALARS) else text
    except Exception:
        low = text.lower()
        if low == "true":
            return True
        if low == "false":
            return False
        return text


class Configurator(object):
    def __init__(self, config_file, default_section="default", argv=None):
        if not os.path.isfile(config_file):
            raise FileNotFoundError("There is not config file named '%s'!" % config_file)
        self._default_section = default_section
        self.cmd_arg = self._read_cmd_arg(sys.argv if argv is None else argv)
        self.lib_arg = self._read_config_file(config_file)
        self.lib_arg["proj_path"] = os.path.dirname(config_file) + "/"
        arg_file = os.path.join(self.lib_arg["config_dir"], self.lib_arg["recommender"] + ".properties")
        if not os.path.isfile(arg_file):   # also accept a conf dir next to the library file
            alt = os.path.join(os.path.dirname(os.path.abspath(config_file)), arg_file)
            arg_file = alt if os.path.isfile(alt) else arg_file
        self.alg_arg = self._read_config_file(arg_file)

    @staticmethod
    def _read_cmd_arg(argv):
        cmd = OrderedDict()
        if argv and "ipykernel_launcher" in argv[0]:
            return cmd
        for token in argv[1:]:
            if not token.startswith("--"):
                raise SyntaxError("Commend arg must start with '--', but '%s' is not!" % token)
            name, value = token[2:].split("=")     # ValueError unless exactly one '='
            cmd[name] = value
        return cmd

    def _read_config_file(self, filename):
        parser = ConfigParser()
        parser.optionxform = str
        parser.read(filename, encoding="utf-8")
        sections = parser.sections()
        if not sections:
            raise ValueError("'%s' is empty!" % filename)
        if len(sections) == 1:
            section = sections[0]
        elif self._default_section in sections:
            section = self._default_section
        else:
            raise ValueError("'%s' has more than one sections but there is no section named '%s'"
                             % (filename, self._default_section))
        args = OrderedDict(parser[section].items())
        for name, value in self.cmd_arg.items():
            if name in args:
                args[name] = value
        return args

    def params_str(self):
        bad = set('/\\":*?<>|\t')
        text = "_".join("%s=%s" % (k, v) for k, v in self.alg_arg.items() if len(v) < 20)
        return "".join("_" if ch in bad else ch for ch in text)

    def __getitem__(self, item):
        if not isinstance(item, str):
            raise TypeError("index must be a str")
        for table in (self.lib_arg, self.alg_arg, self.cmd_arg):
            if item in table:
                return _parse_value(table[item])
        raise KeyError("There are not the parameter named '%s'" % item)

    def __getattr__(self, item):
        if item.startswith("_") or item in ("cmd_arg", "lib_arg", "alg_arg"):
            raise AttributeError(item)
        return self[item]

    def __contains__(self, item):
        return item in self.lib_arg or item in self.alg_arg or item in self.cmd_arg

    def __str__(self):
        lib = "\n".join("%s=%s" % kv for kv in self.lib_arg.items())
        alg = "\n".join("%s=%s" % kv for kv in self.alg_arg.items())
        return "\n\nNeuRec hyperparameters:\n%s\n\n%s's hyperparameters:\n%s\n" % (lib, self["recommender"], alg)

    __repr__ = __str__
