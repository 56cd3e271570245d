"""Benchmark harness around the hot path: PyTorch/MIOpen ResNet-FPN backbones
(SURVEY.md 2.3 marks them out of scope as hand kernels; they exist so that
images/s is measured on the configuration BASELINE.json names).  Not part of
the drop-in product."""
