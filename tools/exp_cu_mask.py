"""Do hipGraph kernel nodes keep the CU mask of the stream they were captured on?  

256 workgroups of 1024 threads, each busy for 50 us: on all 256 CUs one round (~50 us), on a stream masked to 32 CUs eight
rounds (~400 us).  Measured eagerly on the masked stream, then captured on that stream (forked from the capture stream, as the
rulebook / weight-gradient side streams are) and replayed.  Also: which XCDs the workgroups ran on.
usage: python tools/exp_cu_mask.py > profiles/r05_cu_mask.txt"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from com_amd import _lib as L

lib = L.lib()
dev = torch.device('cuda')
TICKS = 5000          # 50 us at 100 MHz


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    out = ctypes.c_void_p()
    L.check(lib.pcd_debug_stream_create_cu_mask(words, 8, ctypes.byref(out)), "cu mask stream")
    return torch.cuda.ExternalStream(out.value, device=dev)


def spin(stream, seen):
    L.check(lib.pcd_debug_spin(256, TICKS, L.ptr(seen), ctypes.c_void_p(stream.cuda_stream)), "spin")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, bits in (("all 256 CUs", (1 << 256) - 1), ("bits 0..31 (32 CUs)", (1 << 32) - 1),
                   ("bits 224..255 (32 CUs)", ((1 << 32) - 1) << 224), ("bits 0..223 (224 CUs)", (1 << 224) - 1)):
    st = masked_stream(bits)
    seen = torch.zeros(1, dtype=torch.int32, device=dev)
    cur = torch.cuda.current_stream()

    def eager():
        st.wait_stream(cur)
        spin(st, seen)
        cur.wait_stream(st)
    t_eager = timed(eager)
    xcc_eager = int(seen.item()) & 0xFFFF
    seen.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        c = torch.cuda.current_stream()
        st.wait_stream(c)
        spin(st, seen)
        c.wait_stream(st)
    t_graph = timed(g.replay)
    xcc_graph = int(seen.item()) & 0xFFFF
    # captured DIRECTLY on the masked stream (the capture stream itself carries the mask)
    seen.zero_()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=st):
        spin(torch.cuda.current_stream(), seen)
    with torch.cuda.stream(st):
        t_graph2 = timed(g2.replay)
    print(f"{name:26s}: eager on the masked stream {t_eager:7.1f} us (XCDs {xcc_eager:#06x}) | forked branch of a graph, replayed "
          f"{t_graph:7.1f} us (XCDs {xcc_graph:#06x}) | graph captured on AND launched into the masked stream {t_graph2:7.1f} us "
          f"(XCDs {int(seen.item()) & 0xFFFF:#06x})", flush=True)
