"""Can a captured hipGraph wait on an EXTERNAL event (recorded by an eager stream before every replay)?"""
import torch, time
dev = "cuda"
a = torch.zeros(1 << 20, device=dev)
b = torch.zeros(1 << 20, device=dev)
ev = torch.cuda.Event(external=True)
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream()
with torch.cuda.stream(side):
    a.fill_(1.0)
    ev.record(side)
torch.cuda.synchronize()
cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)                 # external wait node
        b.copy_(a * 2)
torch.cuda.current_stream().wait_stream(cap)
for k in range(2, 6):
    with torch.cuda.stream(side):
        torch.cuda._sleep(20_000_000)      # make the producer late: the graph must really wait
        a.fill_(float(k))
        ev.record(side)
    g.replay()
    torch.cuda.synchronize()
    print(k, float(b[0]), "ok" if float(b[0]) == 2.0 * k else "WRONG")
