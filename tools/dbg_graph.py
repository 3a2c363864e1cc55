import sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_static as T
from com_amd import hotpath, ops
from com_amd.utils import synth
stage = int(sys.argv[1])
net, bev, batches = T._setup()
w = (torch.randn(2 * 256 * 188 * 188, device='cuda') * 1e-3).bfloat16()
plan = ops.StaticPlan(); ops.PLAN = plan
import os
if "noeager" in os.environ.get("VAR", ""):
    plan.caps["voxels"] = 40000
elif "fwdeager" in os.environ.get("VAR", ""):
    for pts, offs in batches:
        bd = hotpath.transform_points_to_voxels({"points": pts, "frame_offsets": offs, "batch_size": 2}, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, bf16_features=True)
        v = os.environ.get("VAR", "")
        if "E1" in v:
            net.conv1(net.conv_input(net._input_tensor(bd)))
        elif "E2" in v:
            net.conv2(net.conv1(net.conv_input(net._input_tensor(bd))))
        elif "E3" in v:
            net.conv3(net.conv2(net.conv1(net.conv_input(net._input_tensor(bd)))))
        elif "E4" in v:
            net.conv4(net.conv3(net.conv2(net.conv1(net.conv_input(net._input_tensor(bd))))))
        elif "E5" in v:
            net.conv_out(net.conv4(net.conv3(net.conv2(net.conv1(net.conv_input(net._input_tensor(bd)))))))
        elif "E6" in v:
            net(bd)
        elif "E7" in v:
            net._bump_bn_counters(); net.conv2(net.conv1(net.conv_input(net._input_tensor(bd))))
        elif "E0" in v:
            pass
        else:
            bev(net(bd))
else:
    for pts, offs in batches: T._step(net, bev, pts, offs, 2, w)
T._step.last = None
bd = None
plan.active = True
s_pts, s_offs = batches[0][0].clone(), batches[0][1].clone()
import os
VAR = os.environ.get("VAR", "")
_bd0 = hotpath.transform_points_to_voxels({"points": s_pts, "frame_offsets": s_offs, "batch_size": 2}, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, bf16_features=True)
def part():
    if "novox" in VAR:
        bd = dict(_bd0)
    else:
        bd = {"points": s_pts, "frame_offsets": s_offs, "batch_size": 2}
        bd = hotpath.transform_points_to_voxels(bd, synth.WAYMO_RANGE, synth.WAYMO_VOXEL, 5, 150000, bf16_features=True)
    if stage == 1: return
    if "only1" in VAR:
        x = net.conv_input(net._input_tensor(bd)); x = net.conv1(x)
        for p in net.parameters(): p.grad = None
        x.features.float().sum().backward()
        return
    if "only2" in VAR:
        x = net.conv_input(net._input_tensor(bd)); x = net.conv1(x); x = net.conv2(x)
        for p in net.parameters(): p.grad = None
        x.features.float().sum().backward()
        return
    bd = bev(net(bd))
    sf = bd["spatial_features"]
    if stage == 2: return
    if stage >= 6:
        key = {6: "x_conv1", 7: "x_conv2", 8: "x_conv4"}.get(stage)
        t = bd["multi_scale_3d_features"][key] if key else bd["encoded_spconv_tensor"]
        for p in net.parameters(): p.grad = None
        t.features.float().sum().backward() if stage != 10 else None
        return
    loss = (sf.reshape(-1) * w).sum() if stage != 5 else torch.dot(sf.reshape(-1), w)
    if stage == 3: return
    for p in net.parameters(): p.grad = None
    loss.backward()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): part()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print('warm ok', flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    part()
print('captured', flush=True)
g.replay(); torch.cuda.synchronize(); print('replayed', flush=True)
g.replay(); torch.cuda.synchronize(); print('replayed2', plan.check(), flush=True)
for i in range(8):
    g.replay(); torch.cuda.synchronize(); print('replay', i + 3, flush=True)
