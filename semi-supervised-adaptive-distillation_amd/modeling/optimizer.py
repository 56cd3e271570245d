"""Gradient exchange and parameter update operators of a training net, as
detectron/lib/modeling/optimizer.py:33-130 adds them after the backward pass:

  * `NCCLAllreduce(grad, grad)` per parameter (optimizer.py:72-92).  The reference owns all GPUs in one
    process and lists every replica's gradient blob in one operator; here a replica is a process, so the
    operator lists this rank's blob and runs over the process communicator (csrc/ops/nccl_ops.cc,
    workspace.CommInit) -- a no-op without one, as the reference's single-GPU case is (it adds none).
  * per parameter: biases `Scale(grad, grad, scale=2.0)` (no weight decay, 2x learning rate), weights
    `WeightedSum([grad, one, param, wd], grad)`, then `MomentumSGDUpdate([grad, momentum, lr, param],
    [grad, momentum, param], momentum=...)` (optimizer.py:95-130).

`lr`, `one`, `wd` and the `<param>_momentum` blobs are what the reference's param_init_net creates
(ConstantFill); `init_blobs` returns their values for the caller to feed.
"""
import numpy as np


def add_allreduce_ops(model, param_to_grad, params=None):
    """optimizer.py:72-92 for one replica: one in-place NCCLAllreduce per parameter gradient."""
    for p in (params if params is not None else trainable_params(model)):
        g = param_to_grad[p]
        model.net.NCCLAllreduce([g], [g])


def trainable_params(model):
    """model.TrainableParams(): weights and biases in creation order (detector.py:66-90)."""
    return [name for name, _, _ in model.params]


def add_param_update_ops(model, param_to_grad, momentum=0.9, weight_decay=1e-4, params=None):
    """optimizer.py:95-130.  Returns {blob: numpy value} of the param_init_net blobs."""
    init = {"lr": np.zeros(1, np.float32), "one": np.ones(1, np.float32),
            "wd": np.full(1, weight_decay, np.float32)}
    shapes = {name: shape for name, shape, _ in model.params}
    for p in (params if params is not None else trainable_params(model)):
        g = param_to_grad[p]
        init[p + "_momentum"] = np.zeros(shapes[p], np.float32)
        if p in model.biases:
            model.net.Scale([g], [g], scale=2.0)
        elif weight_decay > 0:
            model.net.WeightedSum([g, "one", p, "wd"], [g])
        model.net.MomentumSGDUpdate([g, p + "_momentum", "lr", p], [g, p + "_momentum", p],
                                    momentum=momentum)
    return init
