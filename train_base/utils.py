"""`initialize_module` -- the reference's plug-in loader (train_base/utils.py:68-100): import a
dotted path and instantiate it with `args`.  This is the drop-in boundary's Python face."""
import importlib


def initialize_module(path: str, args: dict = None, initialize: bool = True):
    module_path, _, name = path.rpartition(".")
    obj = getattr(importlib.import_module(module_path), name)
    if not initialize:
        return obj
    return obj(**args) if args else obj()
