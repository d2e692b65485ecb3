"""MI355X-native DaNet hot path (HIP kernels behind the reference's Python call signatures).

Importable as ``danet_densepose2smpl_amd`` (a symlink to this directory; the directory name
itself is not a Python identifier)."""
from . import assets, constants  # noqa: F401


def _lazy(name):
    import importlib
    return importlib.import_module('.' + name, __name__)


def __getattr__(name):
    if name in ('ops', 'smpl', 'renderer', '_lib', 'geometry', 'conv', 'nn', 'models', 'distributed', 'trainer'):
        return _lazy(name)
    if name == 'SMPL':
        return _lazy('smpl').SMPL
    if name == 'IUV_Renderer':
        return _lazy('renderer').IUV_Renderer
    raise AttributeError(name)
