"""MI355X-native hot path of Semi-supervised-Adaptive-Distillation:
SigmoidAdaptiveDistillLoss(+Gradient), PowSum and the RetinaNet subnet
conv3x3 forward/backward behind a Caffe2-shaped Operator<HIPContext> surface.

Import as `ssad_amd` (see ssad_amd.py at the repository root)."""
from . import synth  # noqa: F401
