import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import ada_mvs_amd
from ada_mvs_amd import hip_ops, packing
N, D, h, w = 128, 192, 96, 192
g = torch.Generator().manual_seed(0)
x = torch.randn(N, h * w, D, generator=g).cuda()
wt = torch.randn(D, D, 3, 3, generator=g) / (3 * D ** 0.5)
pk = packing.pack_reg_layer_bf16x3(wt, torch.ones(D), torch.zeros(D), False).cuda()
nw = pk.numel() - D
out = torch.zeros(N, h * w, D, device="cuda")
hip_ops.conv3x3_dd(x, pk[:nw], pk[nw:], None, N, D, h, w, 0, 1, out=out, precision=1)
torch.cuda.synchronize()
nb = N * (h // 8) * (w // 16)
v = out.reshape(-1)[:nb * 4].reshape(nb, 4).cpu()
lds = v[:, 0].long(); hw = v[:, 1].long()
import collections
print("LDS_ALLOC (16 bits) values:", collections.Counter((lds & 0xff).tolist()).most_common(8), "size field:", collections.Counter(((lds >> 12) & 0x1ff).tolist()).most_common(4))
print("HW_ID wave_id:", collections.Counter((hw & 0xf).tolist()).most_common(8), "simd:", collections.Counter(((hw >> 4) & 3).tolist()).most_common(4), "cu:", len(set(((hw >> 8) & 0xf).tolist())), "se/xcc bits sample:", [hex(int(t)) for t in hw[:6]])
