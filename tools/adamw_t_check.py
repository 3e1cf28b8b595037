"""a3v_adamw_scaled_t (AdamW that also writes the transposed bf16 image) against a3v_adamw_scaled + a transpose: bit equality and time per call.
Run on the GPU box: python tools/adamw_t_check.py"""
import torch, sys
sys.path.insert(0, "/root/repo")
from a3vlm_amd import lib as _l, ops
lib = _l.load()
dev = "cuda"
torch.manual_seed(0)
for rows, cols, row0, Np in [(4096, 4096, 0, 4096), (1024, 4096, 4096, 12288), (11008, 4096, 11008, 22016), (4096, 11008, 0, 4096 + 64), (4096, 4096, 0, 4096 + 64), (128, 64, 64, 256)]:
    p = torch.randn(rows, cols, device=dev); g = torch.randn(rows, cols, device=dev) * 0.1
    m = torch.randn(rows, cols, device=dev) * 0.01; v = torch.rand(rows, cols, device=dev) * 1e-3
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    img = torch.zeros(rows, cols, device=dev, dtype=torch.bfloat16); img2 = torch.zeros_like(img)
    wt = torch.zeros(cols, Np, device=dev, dtype=torch.bfloat16)
    gs = torch.tensor([0.7], device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.a3v_adamw_scaled(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 1e-3, 0.9, 0.95, 1e-8, 0.02, 3, img.data_ptr(), gs.data_ptr(), st)
    assert rc == 0
    view = wt[:, row0:row0 + rows]
    rc = lib.a3v_adamw_scaled_t(p2.data_ptr(), g.data_ptr(), m2.data_ptr(), v2.data_ptr(), rows, cols, 1e-3, 0.9, 0.95, 1e-8, 0.02, 3, img2.data_ptr(), view.data_ptr(), wt.stride(0), gs.data_ptr(), st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    ok = torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2) and torch.equal(img, img2) and torch.equal(view, img.t())
    other = wt.clone(); other[:, row0:row0 + rows] = 0
    print(rows, cols, "equal", ok, "untouched elsewhere", bool((other == 0).all()))
    def t_us(fn, n=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    a = t_us(lambda: lib.a3v_adamw_scaled(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 1e-3, 0.9, 0.95, 1e-8, 0.02, 3, img.data_ptr(), gs.data_ptr(), st))
    b = t_us(lambda: lib.a3v_adamw_scaled_t(p2.data_ptr(), g.data_ptr(), m2.data_ptr(), v2.data_ptr(), rows, cols, 1e-3, 0.9, 0.95, 1e-8, 0.02, 3, img2.data_ptr(), view.data_ptr(), wt.stride(0), gs.data_ptr(), st))
    n = rows * cols
    print(f"   adamw {a:.1f} us ({30*n/a/1e6:.2f} TB/s)   adamw_t {b:.1f} us ({32*n/b/1e6:.2f} TB/s)")
