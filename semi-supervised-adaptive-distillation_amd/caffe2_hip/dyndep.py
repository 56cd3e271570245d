"""caffe2.python.dyndep (caffe2/python/dyndep.py:28-70): load an operator
library and refresh the registered-operator list.  Detectron calls
`dyndep.InitOpsLibrary(path_to_libcaffe2_detectron_ops_gpu.so)`
(detectron/lib/utils/c2.py:39-42); here the library is
libcaffe2_detectron_ops_hip.so."""
from . import _capi, core

_loaded = set()


def InitOpsLibrary(name=None):
    _capi.load(name)
    _loaded.add(name or _capi.LIB_PATH)
    core.RefreshRegisteredOperators()


def GetImportedOpsLibraries():
    return set(_loaded)
