"""Builder tool (GPU box): what this chip's HBM delivers to the plainest kernels, beside the narrow decoder stages' 8 B per element (read a plane, write a plane).
Times torch's own elementwise kernels on planes of the decoder's size (0.94 GB = one C = 16 stage plane at B = 32) with HIP events: copy (1 read + 1 write),
read-only (sum), write-only (fill), and a 3-operand add (2 reads + 1 write).  Prints one JSON line per kernel."""
import json, torch
assert torch.cuda.is_available()
n = 459264 * 32 * 16      # f32 elements of the last stage's plane at B = 32
x = torch.randn(n, device="cuda"); y = torch.empty_like(x); z = torch.randn(n, device="cuda")
def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, f, nbytes in (("copy (read + write)", lambda: y.copy_(x), 8 * n), ("read only (sum)", lambda: x.sum(), 4 * n), ("write only (fill)", lambda: y.fill_(1.0), 4 * n),
                        ("add (2 reads + 1 write)", lambda: torch.add(x, z, out=y), 12 * n), ("leaky_relu (read + write)", lambda: torch.nn.functional.leaky_relu(x, 0.1, inplace=False), 8 * n)):
    ms = timed(f)
    print(json.dumps({"kernel": name, "plane_GB": round(4 * n / 1e9, 3), "ms": round(ms, 4), "TB_per_s": round(nbytes / ms / 1e9, 3)}), flush=True)
