"""Import helper: the package directory is named ``nodey-audio-editor_amd`` (hyphen, mirroring the upstream
repository name), which Python cannot import by name.  ``load()`` registers it as ``nodey_audio_editor_amd``."""
import importlib.util
import os
import sys

_NAME = "nodey_audio_editor_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nodey-audio-editor_amd")


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_DIR, "__init__.py"),
                                                  submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
