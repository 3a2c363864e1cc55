"""Is the H2D rate of pinned buffers a NUMA matter on this box?  Prints the GPU's PCI locality and copies 64 MB pinned
buffers allocated under different CPU affinities."""
import os, glob, time, torch
p = torch.cuda.get_device_properties(0)
bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
base = f"/sys/bus/pci/devices/{bdf}"
def rd(f):
    try:
        return open(os.path.join(base, f)).read().strip()
    except Exception as e:
        return f"<{type(e).__name__}>"
print("gpu", bdf, "numa_node", rd("numa_node"), "local_cpulist", rd("local_cpulist"))
print("affinity", len(os.sched_getaffinity(0)), "cpus; nodes:", [os.path.basename(d) for d in glob.glob("/sys/devices/system/node/node*")])
for n in sorted(glob.glob("/sys/devices/system/node/node*")):
    print(os.path.basename(n), open(n + "/cpulist").read().strip())
def parse(s):
    out = set()
    for part in s.split(","):
        if "-" in part:
            a, b = part.split("-"); out |= set(range(int(a), int(b) + 1))
        elif part.strip():
            out.add(int(part))
    return out
def bw(tag):
    h = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
    h.fill_(1)
    d = torch.empty_like(h, device="cuda")
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(tag, f"{64 / 1024 / dt:.1f} GB/s")
allc = sorted(os.sched_getaffinity(0))
bw("default affinity")
for n in sorted(glob.glob("/sys/devices/system/node/node*")):
    cpus = parse(open(n + "/cpulist").read().strip()) & set(allc)
    if cpus:
        os.sched_setaffinity(0, cpus)
        bw("bound to " + os.path.basename(n))
os.sched_setaffinity(0, set(allc))
