"""hipGraph replay for the small, launch-bound model forwards around the dense lookup.

A batch-1 query embedding or a 16-pair rerank batch of a 0.6 B model is ~450 tiny kernels; in
eager PyTorch the GPU waits for the launches. `GraphRunner` captures the forward once per
(batch, padded length) bucket into a `torch.cuda.CUDAGraph` (a hipGraph on ROCm) and replays it:
inputs are left-padded to the bucket (the clients already pool / read logits at the LAST position,
and RoPE is relative, so left padding does not change the result), batch rows are padded by
repeating row 0. Anything that cannot be captured falls back to the eager call, loudly.
"""

from __future__ import annotations

import logging

logger = logging.getLogger(__name__)


class GraphRunner:
    def __init__(self, fn, seq_step: int = 32, batch_step: int = 1, enabled: bool = True):
        self.fn, self.seq_step, self.batch_step, self.enabled = fn, seq_step, batch_step, enabled
        self._graphs: dict = {}
        self.replays = 0

    def __call__(self, input_ids, attention_mask):
        import torch

        if not (self.enabled and input_ids.is_cuda):
            return self.fn(input_ids, attention_mask)
        b, length = input_ids.shape
        L = -(-length // self.seq_step) * self.seq_step
        B = -(-b // self.batch_step) * self.batch_step
        key = (B, L)
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._capture(B, L, input_ids, attention_mask)
            self._graphs[key] = entry
        if entry is False:
            return self.fn(input_ids, attention_mask)
        graph, s_ids, s_mask, s_out = entry
        s_ids.zero_()
        s_mask.zero_()
        s_ids[:b, L - length:] = input_ids
        s_mask[:b, L - length:] = attention_mask
        if B > b:  # filler rows: copies of row 0 (a fully masked row would produce NaNs)
            s_ids[b:] = s_ids[0]
            s_mask[b:] = s_mask[0]
        graph.replay()
        self.replays += 1
        return s_out[:b].clone()

    def _capture(self, B, L, input_ids, attention_mask):
        import torch

        dev = input_ids.device
        s_ids = torch.ones((B, L), dtype=input_ids.dtype, device=dev)
        s_mask = torch.ones((B, L), dtype=attention_mask.dtype, device=dev)
        s_mask[:, : L // 2] = 0  # capture WITH padding so no "mask is all ones" shortcut is baked in
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.fn(s_ids, s_mask)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                s_out = self.fn(s_ids, s_mask)
            return graph, s_ids, s_mask, s_out
        except Exception as exc:  # noqa: BLE001 - any capture failure means "run eagerly"
            logger.warning("hipGraph capture failed for bucket %s (%s); running eagerly", (B, L), exc)
            torch.cuda.synchronize(dev)
            return False
