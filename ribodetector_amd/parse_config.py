"""ConfigParser - the config.json surface of the reference (parse_config.py:10-92, utils/util.py:13-16):
`from_json(path)`, `cfg['key']`, `cfg.init_obj('arch', module)`, `cfg.init_ftn(...)`, `cfg.get_logger(...)`.
Same keys and semantics (`arch.type`, `arch.args.*`, `state_file.{mcc,recall}`, `n_gpu`); plus `load_state_dict(key)`,
which resolves `state_file[key]` relative to the config file and reads either a .safetensors file (shipped) or a
reference .pth checkpoint (torch.load(...)['state_dict'], reference detect.py:101,115)."""
import json
import logging
import os
from collections import OrderedDict
from functools import partial


class ConfigParser:
    LOG_LEVELS = {0: logging.WARNING, 1: logging.INFO, 2: logging.DEBUG}

    def __init__(self, config, path=None):
        self.config = config
        self.path = path
        self.log_levels = dict(self.LOG_LEVELS)

    @classmethod
    def from_json(cls, config_json):
        with open(str(config_json), "rt") as fh:
            return cls(json.load(fh, object_hook=OrderedDict), os.path.abspath(str(config_json)))

    def __getitem__(self, name):
        return self.config[name]

    def _resolve(self, name, kwargs):
        spec = self[name]
        args = dict(spec["args"])
        clash = [k for k in kwargs if k in args]
        assert not clash, "Overwriting kwargs given in config file is not allowed"
        args.update(kwargs)
        return spec["type"], args

    def init_obj(self, name, module, *args, **kwargs):
        """`module.<config[name]['type']>(*args, **config[name]['args'], **kwargs)`"""
        typ, margs = self._resolve(name, kwargs)
        return getattr(module, typ)(*args, **margs)

    def init_ftn(self, name, module, *args, **kwargs):
        typ, margs = self._resolve(name, kwargs)
        return partial(getattr(module, typ), *args, **margs)

    def get_logger(self, name, verbosity=2, logfile=None):
        assert verbosity in self.log_levels, "verbosity option {} is invalid. Valid options are {}.".format(
            verbosity, self.log_levels.keys())
        handlers = [logging.StreamHandler()]
        if logfile is not None:
            handlers.append(logging.FileHandler(logfile, mode="w"))
        logging.basicConfig(level=self.log_levels[verbosity], format="%(asctime)s : %(levelname)s  %(message)s",
                            datefmt="%Y-%m-%d %H:%M:%S", handlers=handlers)
        return logging.getLogger(name)

    # ---- weights lookup ------------------------------------------------------------------------------
    def state_file(self, key="mcc"):
        rel = self["state_file"][key]
        base = os.path.dirname(self.path) if self.path else os.path.dirname(os.path.abspath(__file__))
        return rel if os.path.isabs(rel) else os.path.join(base, rel)

    def load_state_dict(self, key="mcc"):
        path = self.state_file(key)
        if path.endswith(".safetensors"):
            from safetensors.numpy import load_file
            return load_file(path)
        import torch
        state = torch.load(path, map_location="cpu")
        return state["state_dict"] if "state_dict" in state else state
