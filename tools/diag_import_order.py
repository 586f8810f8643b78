import faulthandler, sys
faulthandler.enable()
order = sys.argv[1]
if order == "torch_first":
    import torch
    print("torch imported", flush=True)
    import stark_anatomy_amd as sa
else:
    import stark_anatomy_amd as sa
    print("sa imported", flush=True)
    import torch
print("both imported", flush=True)
maps = open('/proc/self/maps').read()
print(sorted({l.split()[-1] for l in maps.splitlines() if 'amdhip' in l or 'hsa-runtime' in l}), flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "torchstream":
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    print("torch stream", s.cuda_stream, flush=True)
    ctx = sa.Context(0, stream=s.cuda_stream)
else:
    ctx = sa.Context(0)
print("ctx ok", flush=True)
import numpy as np
x = np.arange(2 * 1024, dtype=np.uint64).reshape(-1, 2)
y = sa.ntt(x, ctx=ctx)
print("ntt ok", y[:2], flush=True)
t = torch.ones(1024, device="cuda") * 2
torch.cuda.synchronize()
print("torch ok", float(t.sum()), flush=True)
