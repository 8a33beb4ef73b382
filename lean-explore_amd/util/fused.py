"""Fewer, larger kernels for the Qwen3 forwards around the dense lookup (PyTorch-ROCm plumbing).

HF's eager `Qwen3RMSNorm.forward` is six elementwise / reduction kernels (cast, pow, mean, add eps,
rsqrt, two multiplies); a 28-layer model calls it 113 times per forward, and a rocprofv3 trace of
config 5 shows those pieces at ~30 % of the models' GPU time. `fuse_rmsnorm` re-points every
RMSNorm module of a loaded model at `torch.nn.functional.rms_norm` (one kernel, float32
accumulation for half inputs like the eager code). Same mathematics; the weight multiply is done
before instead of after the cast back to the input dtype, i.e. results agree to the input dtype's
rounding (tests/test_model_clients.py)."""

from __future__ import annotations


def fuse_rmsnorm(model) -> int:
    """Returns the number of modules switched. No-op when torch lacks F.rms_norm."""
    import types

    import torch.nn.functional as F

    if not hasattr(F, "rms_norm"):
        return 0
    n = 0
    for m in model.modules():
        if type(m).__name__.endswith("RMSNorm") and hasattr(m, "weight") and hasattr(m, "variance_epsilon"):
            def forward(self, hidden_states):
                return F.rms_norm(hidden_states, (hidden_states.shape[-1],), self.weight,
                                  self.variance_epsilon)

            m.forward = types.MethodType(forward, m)
            n += 1
    return n
