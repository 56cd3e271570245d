"""Host-side mirror of the slice of `caffe2.python` the hot path is driven
through in the reference: `core` (CreateOperator, Net, DeviceOption,
gradient generation), `workspace` (FeedBlob / FetchBlob / RunOperatorOnce /
CreateNet / RunNet) and `dyndep` (InitOpsLibrary).  Same names, argument
meaning and error behaviour; underneath, everything goes through the C-ABI of
libcaffe2_detectron_ops_hip.so (include/c2hip_capi.h)."""
from . import caffe2_pb2, core, dyndep, workspace  # noqa: F401
