"""hipGraph capture of a whole training step (torch.cuda.CUDAGraph drives hipStreamBeginCapture on ROCm).

The C-ABI kernels are launched on torch's current stream, so they are recorded like any other work; the weight-gradient
side stream forks from and joins the capturing stream through events and becomes part of the graph.  What a step must
satisfy to be capturable: static shapes and device-resident inputs (CTC targets / lengths as CUDA tensors with
`CTCLoss.max_target_length` set), `FusedAdam(capturable=True)` (step count on the device), no `.item()` / `.cpu()`.
At B >= 512 the step is GPU-bound and a graph buys nothing; at the reference's real batch sizes (tens of strips) the
step is bound by ~11 us of host time per launch and the replay removes it.
"""
import torch


class GraphedStep:
    def __init__(self, fn, warmup=3):
        """fn(): one full step on static tensors; its return value (tensor or tuple of tensors) is static too."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm-up off the default stream, as torch's graph recipe asks
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        from . import ops
        self.graph = torch.cuda.CUDAGraph()
        ops.CAPTURE["token"] = object()                  # derived weight forms: built once per weight epoch inside the graph
        try:
            with torch.cuda.graph(self.graph):
                self.out = fn()
        finally:
            ops.CAPTURE["token"] = None

    def __call__(self):
        self.graph.replay()
        # a captured FusedAdam writes the weights through raw pointers and runs no Python on replay, so neither the tensors'
        # _version nor the optimiser's epoch bump moves: every derived weight form cached by an EARLIER eager pass
        # (qea.ops.weight_cached: dgrad filters, planes, W_hh packs) is stale from here on
        from . import ops
        ops.bump_weight_epoch()
        return self.out
