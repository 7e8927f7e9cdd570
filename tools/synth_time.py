import importlib, sys
sys.path.insert(0, ".")
import torch
pkg = importlib.import_module("libperseus-sdr_amd")
dev = torch.device("cuda", 0)
n = 6 << 28
buf = torch.empty(n, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
L = pkg.ddc_lib()
for _ in range(5):
    pkg.check(L.pddc_synth_lcg(buf.data_ptr(), n, 12345, 0, st))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    pkg.check(L.pddc_synth_lcg(buf.data_ptr(), n, 12345, 0, st))
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"k_synth_lcg 6*2^28 bytes: {ms:.4f} ms = {n / ms / 1e9:.2f} TB/s written")
