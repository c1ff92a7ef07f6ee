"""Times one SLSTM layer's recurrence and BPTT: resident kernels (lstm_persist.hip; fp32, and bf16 x 3 for 17 .. 32 columns) vs one
launch per step (lstm.hip).  FAC_LSTM_PERSIST_MAX_BATCH=32 times the fp32 resident kernel at 32 columns too."""
import json
import sys

import torch

sys.path.insert(0, ".")
from facodec_amd import ops  # noqa: E402


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda:0")
    out = []
    for B, H, T in ((16, 1024, 160), (16, 1536, 160), (32, 1024, 160), (32, 1536, 160)):
        BP = 32 * ((B + 31) // 32)
        pre = torch.zeros(4 * H, T, BP, device=dev)
        pre[:, :, :B] = torch.randn(4 * H, T, B, device=dev)
        w = (torch.rand(4 * H, H, device=dev) * 2 - 1) / H ** 0.5
        gates, cs = torch.empty(4 * H, T, BP, device=dev), torch.empty(H, T, BP, device=dev)
        d_out = torch.zeros(H, T, BP, device=dev)
        d_out[:, :, :B] = torch.randn(H, T, B, device=dev)
        whh = ops.pack_lstm_whh(w)
        row = dict(B=B, H=H, T=T)
        row["fwd_per_step_ms"] = timed(lambda: ops.lstm_layer(pre, whh, H, save=(gates, cs)))
        row["bwd_per_step_ms"] = timed(lambda: ops.lstm_layer_bwd(d_out, w, gates, cs, H))
        if ops.lstm_persist_ok(H, B):
            row["fwd_resident_ms"] = timed(lambda: ops.lstm_layer_persist(pre, w, H, B, save=(gates, cs)))
            row["bwd_resident_ms"] = timed(lambda: ops.lstm_layer_bwd(d_out, w, gates, cs, H, batch=B))
            row["fwd_resident_us_per_step"] = 1e3 * row["fwd_resident_ms"] / T
            row["bwd_resident_us_per_step"] = 1e3 * row["bwd_resident_ms"] / T
        if ops.lstm_persist_split_ok(H, B):
            row["fwd_infer_per_step_ms"] = timed(lambda: ops.lstm_layer(pre, whh, H))
            row["fwd_split_resident_ms"] = timed(lambda: ops.lstm_layer_persist_split(pre, w, H, B))
            row["fwd_split_resident_us_per_step"] = 1e3 * row["fwd_split_resident_ms"] / T
            row["split_vs_per_step_rel"] = float((ops.lstm_layer_persist_split(pre, w, H, B)[:, :, :B] - ops.lstm_layer(pre, whh, H)[:, :, :B]).abs().max())
        out.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
