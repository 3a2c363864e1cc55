"""VoxelBackBone8x / VoxelResBackBone8x: the layer graphs of
pcdet/models/backbones_3d/spconv_backbone.py:69-180 and :183-293 (state-dict names per
SURVEY.md Appendix B) over com_amd.spconv."""
from functools import partial

import torch
from torch import nn

from .. import spconv
from ..spconv import SparseConvTensor
from ..spconv import functional as Fsp


def replace_feature(out, new_features):
    """pcdet/utils/spconv_utils.py:28-34"""
    return out.replace_feature(new_features)


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0,
                   conv_type='subm', norm_fn=None):
    """spconv_backbone.py:8-27"""
    if conv_type == 'subm':
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                   bias=False, indice_key=indice_key)
    elif conv_type == 'inverseconv':
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key,
                                          bias=False)
    else:
        raise NotImplementedError
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class SparseBasicBlock(spconv.SparseModule):
    """spconv_backbone.py:30-66 (both convs have bias=True: `bias = norm_fn is not None`)."""
    expansion = 1
    fuse_identity_grad = True      # identity-branch gradient added inside conv1's dgrad kernel

    def __init__(self, inplanes, planes, stride=1, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        assert norm_fn is not None
        bias = norm_fn is not None
        self.conv1 = spconv.SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=bias,
                                       indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=bias,
                                       indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # same dataflow as spconv_backbone.py:50-66; bn -> relu and bn -> (+identity) -> relu each run as
        # one fused stats pass + one fused apply pass (com_amd/csrc/fused.hip)
        # the identity branch leaves conv1 as a second output, so its gradient is added in conv1's dgrad epilogue
        fuse_identity = (self.fuse_identity_grad and self.downsample is None and x.features.is_cuda
                         and torch.is_grad_enabled())
        if fuse_identity:
            out, identity_features = self.conv1(x, passthrough=True)
        else:
            out = self.conv1(x)
            identity_features = (self.downsample(x) if self.downsample is not None else x).features
        out = replace_feature(out, Fsp.batch_norm_act(self.bn1, out.features, None, True, out.num_rows))
        out = self.conv2(out)
        out = replace_feature(out, Fsp.batch_norm_act(self.bn2, out.features, identity_features, True,
                                                      out.num_rows))
        return out


class _RulebookPrefetcher:
    event_per_rulebook = False     # (one event per rulebook instead of per unit -- measured: no difference)

    def __init__(self, units, x0, side):
        self.units, self.t, self.side, self.next = units, x0, side, 0
        self.after_units = x0.indice_dict.pop("__after_units__", None)
        self.hook_at = x0.indice_dict.pop("__after_units_at__", None)

    def advance(self, inline=False, stream=None):
        """inline: build the unit on the CURRENT stream (no event: its consumers are ordered behind it anyway);
        stream: build it on that stream instead of the prefetcher's own."""
        if self.next >= len(self.units):
            return
        unit = self.units[self.next]
        self.next += 1
        side = stream if stream is not None else self.side
        with torch.cuda.stream(torch.cuda.current_stream() if inline else side):
            t = self.t
            built = []
            for conv in unit:
                if conv.subm:
                    t.indice_dict["__subm_hint__"] = self._subm_hint(conv, unit, t)
                rb, out_idx, out_shape = conv._rulebook(t)
                t.indice_dict.pop("__subm_hint__", None)
                if getattr(rb, "ready_event", None) is None:
                    built.append(rb)
                    self._window_plan(conv, rb)
                    self._pair_plan(conv, rb)
                    if not inline and self.event_per_rulebook:
                        # the strided conv that opens a level waits for ITS rulebook only, not for the SubM rulebook (+ window
                        # plan) of the level behind it in the same unit
                        rb.ready_event = torch.cuda.Event()
                        rb.ready_event.record(side)
                if not conv.subm:
                    t = SparseConvTensor(t.features, out_idx, out_shape, t.batch_size, indice_dict=t.indice_dict,
                                         num_rows=rb.n_out_dev)
            self.t = t
            if built and not inline and not self.event_per_rulebook:
                ev = torch.cuda.Event()
                ev.record(side)
                for rb in built:
                    rb.ready_event = ev
            from .. import ops
            if ops.STAMPS is not None:
                ops.stamp(f"rb_unit{self.next - 1}")
            if self.next >= len(self.units) and self.after_units is not None:
                # the last unit is on the stream: nothing issued later reads the voxeliser's outputs (level-1 coordinates, rank
                # map) -- a caller that recycles those buffers continues HERE, on this stream, idle for the rest of the forward
                if self.hook_at is None:
                    self.run_hook()                # (we are on the rulebook stream)


    def run_hook(self, after_stream=None):
        """The caller's after_rulebooks hook, on the rulebook stream behind the last unit -- and, with `after_stream`, behind
        whatever that stream has been given so far (backbone.after_rulebooks_at: the hook's work then starts when the main
        chain has finished that level instead of right behind the rulebook chain)."""
        if self.after_units is None or self.next < len(self.units):
            return
        hook, self.after_units = self.after_units, None
        side = self.side
        with torch.cuda.stream(side):
            u0 = getattr(self, "unit0_stream", None)
            if u0 is not None and u0 is not side:
                # unit 0 (the level-1 SubM rulebook) reads the voxeliser's coordinates and rank map on ITS OWN stream: the
                # hook may overwrite them, so this stream is ordered behind unit 0's reads first (in a captured graph
                # nothing else puts an edge between the two branches)
                side.wait_stream(u0)
            if after_stream is not None:
                side.wait_stream(after_stream)
            hook()

    @staticmethod
    def _subm_hint(conv, unit, t):
        """(window width or None, neighbour table needed) for the SubM rulebook `conv` is about to build: all SubM convs of the
        unit with its kernel share that rulebook (SparseConvolution._rulebook), so the table is only needed if one of them
        runs outside the window kernels -- forward / data gradient (window_capable) or, when autograd will run, the weight
        gradient (option "subm_window_wgrad")."""
        group = [c for c in unit if c.subm and tuple(c.kernel_size) == tuple(conv.kernel_size)
                 and tuple(c.dilation) == tuple(conv.dilation)]
        widths = {c.out_channels for c in group if c.window_capable()}
        if len(widths) != 1:
            return None, True
        w = widths.pop()
        bf16 = t.features.dtype == torch.bfloat16
        tables = not bf16 or any(not c.window_capable() or getattr(c, "fp8_train", None) is not None
                                 or (c._needs_backward(t) and not Fsp._window_wgrad(c.out_channels)) for c in group)
        return w, tables

    @staticmethod
    def _pair_plan(conv, rb):
        """The segment tables of the pair-driven strided kernel (ops.pair_conv_plan caches them on the rulebook), here on the
        rulebook stream instead of in front of the conv that needs them."""
        from .. import ops
        if conv.subm or rb._pairs is None:
            return
        cin = ops.pow2_ge8(conv.in_channels)
        if ops.pair_conv_usable(rb, cin, conv.out_channels):
            ops.pair_conv_plan(rb, 0)
        if ops.pair_conv_usable(rb, conv.out_channels, cin):
            ops.pair_conv_plan(rb, 1)

    @staticmethod
    def _window_plan(conv, rb):
        """The window kernel's tile plan of a freshly built SubM rulebook (ops.subm_window_plan caches it on the rulebook),
        here on the rulebook stream: built by the first conv that uses it, it sat on the main chain (~30 us per level)."""
        from .. import ops, _lib as L
        if not (conv.subm and tuple(conv.kernel_size) == (3, 3, 3) and getattr(rb, "order", None) == ops.ROWS_YXZ):
            return
        c = conv.out_channels
        bit = {64: 1, 32: 2, 16: 4, 128: 8}.get(c, 0)
        if bit and (L.get_option("subm_window") & bit) and ops.subm_window_tile_rows(c, c) > 0:
            ops.subm_window_plan(rb, c, c)


class _BackboneBase(nn.Module):
    feature_dtype = torch.bfloat16

    def _input_tensor(self, batch_dict):
        voxel_features, voxel_coords = batch_dict['voxel_features'], batch_dict['voxel_coords']
        batch_size = batch_dict['batch_size']
        feats = voxel_features
        if self.feature_dtype is not None and feats.dtype != self.feature_dtype:
            feats = feats.to(self.feature_dtype)
        x = SparseConvTensor(features=feats, indices=voxel_coords.int(), spatial_shape=self.sparse_shape,
                             batch_size=batch_size, num_rows=batch_dict.get('voxel_num_rows', None))
        rank = batch_dict.get('voxel_rank', None)
        if rank is not None and rank.indices is x.indices and list(rank.shape) == list(x.spatial_shape):
            # key-ordered voxel rows: the voxeliser's bitmap ranks ARE the row ids -> level-1 SubM without a hash table
            x.indice_dict[("__rank__", x.indices.data_ptr())] = rank
            x.indice_dict["__row_order__"] = rank.order      # the strided builds number their rows the same way
        return x

    def _bump_bn_counters(self):
        """num_batches_tracked += 1 for every BatchNorm1d in ONE multi-tensor launch instead of 21."""
        if not hasattr(self, "_bn_list"):
            self._bn_list = [m for m in self.modules() if isinstance(m, nn.BatchNorm1d)]
            for m in self._bn_list:
                m._defer_nbt = True
        if self.training:
            counters = [m.num_batches_tracked for m in self._bn_list if m.num_batches_tracked is not None]
            if counters:
                torch._foreach_add_(counters, 1)

    prefetch_rulebooks = True
    # Optional callable, run ONCE per forward on the rulebook stream right behind its last unit -- the point from which nothing
    # issued later reads the voxeliser's outputs (level-1 coordinates, coordinate -> row map): a caller that recycles those
    # buffers (com_amd.train.CapturedStep voxelises the NEXT batch into them, inside the captured step) continues there.
    # Preferred form: per forward, batch_dict['after_rulebooks_hook'] (+ 'after_rulebooks_at'); this attribute is the
    # fallback when the batch_dict carries none.  None: nothing runs.
    after_rulebooks = None
    # None: the hook runs right behind the last rulebook unit; "conv2" / "conv3" / "conv4": additionally not before the main
    # chain has finished that level (the hook's kernels then run beside the levels after it)
    after_rulebooks_at = None
    first_unit_inline = False      # measured: see _prefetch_rulebooks
    unit0_own_stream = True
    # rulebook units issued before the first conv; each unit's first consumer issues one more
    prefetch_depth = 2

    def _prefetch_rulebooks(self, x0):
        """All 9 rulebooks depend only on the voxel coordinates, not on features: they are built on a second HIP
        stream, one UNIT (a strided conv's rulebook + the SubM rulebook of the level it opens) ahead of the
        feature kernels, so the latency-bound, LDS-free hash / bitmap kernels run beside the gather-GEMMs.  The
        first conv that consumes a unit waits on the unit's event and then issues the next unit.  The weight
        packs (forward + dgrad) go to a third stream.  Captured as fork/join in hipGraph mode."""
        if not (self.prefetch_rulebooks and x0.features.is_cuda):
            return
        if not hasattr(self, "_conv_list"):
            self._conv_list = [m for m in self.modules() if isinstance(m, spconv.conv.SparseConvolution)]
            units = [[]]
            for conv in self._conv_list:
                if not conv.subm and units[-1]:
                    units.append([])
                units[-1].append(conv)
            self._rb_units = units
        cur = torch.cuda.current_stream()
        dev = x0.features.device
        side = Fsp._side_stream(dev, "rulebook")
        pf = _RulebookPrefetcher(self._rb_units, x0, side)
        x0.indice_dict["__prefetcher__"] = pf
        depth = max(1, int(self.prefetch_depth))
        side.wait_stream(cur)
        if self.first_unit_inline:
            # (experiment, off) the level-1 rulebook on the MAIN stream.  Device-clock stamps show the first conv of the step
            # starting 210-260 us into the forward although its rulebook is ready at 24 us (the hipGraph executor runs it
            # behind the next unit's kernels); built inline it finishes at 63 us and the forward ends 84 us earlier -- but
            # the rulebook units, now beside the gather kernels from the start, finish 90-200 us later and the step is the
            # same: 3.44 vs 3.42 ms without stamps (tools/exp_rb_inline.sh)
            pf.advance(inline=True)
            depth -= 1
        if self.unit0_own_stream and not self.first_unit_inline:
            # The level-1 SubM rulebook (unit 0: all the first conv waits for) on a short branch of its own; the other units do
            # not read it (unit 1 = the strided build over the level-1 coordinates), so they fork from the main stream too.  As
            # children of unit 0's last node they held the first conv back: the graph executor ran it behind the whole of unit 1
            # (first conv at 0.27 ms with its rulebook ready at 0.06).
            u0 = Fsp._side_stream(dev, "rulebook0")
            u0.wait_stream(cur)
            pf.advance(stream=u0)
            pf.unit0_stream = u0
            depth -= 1
        for _ in range(depth):
            pf.advance()
        if getattr(self, "_packed_ahead", False):              # pack_after_update() ran since the last update
            self._packed_ahead = False
            # ... and no weight changed behind torch's back since (load_state_dict / broadcast / manual edits bump
            # `_version`; the flat Adam kernel does not, and it is the update pack_after_update() follows)
            if self._packed_versions == [c.weight._version for c in self._conv_list]:
                return
        pack_side = Fsp._side_stream(dev)                      # the wgrad stream is idle during the forward
        pack_side.wait_stream(cur)
        with torch.cuda.stream(pack_side):
            self._pack_all()
        cur.wait_stream(pack_side)

    def pack_after_update(self):
        """Pack the weights for the NEXT forward / backward now, on the current stream -- call it right after the
        optimizer step: the weights do not change again before the next forward, and the launch then runs beside
        whatever else sits between two steps (bench.py: the voxelisation of the next batch) instead of in front of
        the first conv."""
        if not hasattr(self, "_conv_list"):
            return                                             # no forward yet: the first one packs by itself
        self._pack_all()
        self._packed_ahead = True
        self._packed_versions = [c.weight._version for c in self._conv_list]

    def _pack_all(self):
        """All forward (+ dgrad) weight packs of the backbone in one launch."""
        from .. import ops
        todo = []
        for conv in self._conv_list:
            todo.append((conv, 0))
            if self.training and conv.in_channels >= 16:
                todo.append((conv, 1))
        # (layers that run through the window kernel -- decided by their last forward -- get that kernel's layout: modes 2 / 3)
        wm = [(c.weight.detach(), m + (2 if c.use_window else 0)) for c, m in todo]
        if not all(w.dtype == torch.float32 and w.is_contiguous() for w, _ in wm):
            for conv in self._conv_list:
                conv.prepack(dgrad=self.training)
            return
        plan = getattr(self, "_pack_plan", None)
        if plan is None or not plan.valid_for(wm):
            plan = self._pack_plan = ops.PackPlan(wm)
        packed = plan.run()
        got = {}
        for (conv, mode), buf in zip(todo, packed):
            got.setdefault(conv, [None, None])[mode] = buf
        for conv, (f, d) in got.items():
            conv.adopt_packs(f, d)        # (stamped with the conv's current use_window: a later change of mind repacks)

    def _run(self, batch_dict):
        self._bump_bn_counters()
        from .. import ops
        x0 = self._input_tensor(batch_dict)
        ops.stamp("fwd_begin")
        if ops.STAMPS is not None:
            ops.STAMPS["conv_seq"] = 0
        # the hook of THIS forward: batch_dict['after_rulebooks_hook'] (+ optional 'after_rulebooks_at'), per call -- what
        # com_amd.train.CapturedStep passes; the module attributes below are the fallback for callers that set them once
        hook = batch_dict.pop('after_rulebooks_hook', None)
        hook_at = batch_dict.pop('after_rulebooks_at', self.after_rulebooks_at)
        if hook is None:
            hook, hook_at = self.after_rulebooks, self.after_rulebooks_at
        if hook is not None:
            x0.indice_dict["__after_units__"] = hook
            x0.indice_dict["__after_units_at__"] = hook_at
        self._prefetch_rulebooks(x0)
        x = self.conv_input(x0)
        ops.stamp("conv_input")
        x_conv1 = self.conv1(x)
        ops.stamp("conv1")
        pf_ = x0.indice_dict.get("__prefetcher__", None)

        def hook_point(name):
            if pf_ is not None and pf_.hook_at == name:
                pf_.run_hook(after_stream=torch.cuda.current_stream())
        x_conv2 = self.conv2(x_conv1)
        ops.stamp("conv2")
        hook_point("conv2")
        x_conv3 = self.conv3(x_conv2)
        ops.stamp("conv3")
        hook_point("conv3")
        # (level 4 child by child: a hook point behind its strided conv -- "conv4.0" -- lets the hook's kernels start beside the
        #  128-channel SubM layers instead of beside the strided 64 -> 128 conv, the most contention-sensitive kernel of the chain)
        x4 = x_conv3
        for ci, child in enumerate(self.conv4):
            x4 = child(x4)
            if ci == 0:
                hook_point("conv4.0")
        x_conv4 = x4
        ops.stamp("conv4")
        hook_point("conv4")
        out = self.conv_out(x_conv4)
        ops.stamp("conv_out")
        if "__prefetcher__" in x0.indice_dict:                 # every unit was consumed; join the stream(s) anyway
            pf = x0.indice_dict.pop("__prefetcher__")
            pf.run_hook(after_stream=torch.cuda.current_stream())      # (a hook point that came before the last unit was issued)
            torch.cuda.current_stream().wait_stream(pf.side)
            if getattr(pf, "unit0_stream", None) is not None:
                torch.cuda.current_stream().wait_stream(pf.unit0_stream)
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8})
        batch_dict.update({'multi_scale_3d_features': {
            'x_conv1': x_conv1, 'x_conv2': x_conv2, 'x_conv3': x_conv3, 'x_conv4': x_conv4}})
        batch_dict.update({'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return batch_dict


class VoxelBackBone8x(_BackboneBase):
    """spconv_backbone.py:69-180"""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        g = [int(v) for v in grid_size]
        self.sparse_shape = [g[2] + 1, g[1], g[0]]
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'),
            norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = spconv.SparseSequential(block(16, 16, 3, norm_fn=norm_fn, padding=1, indice_key='subm1'))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'))
        self.conv4 = spconv.SparseSequential(
            block(64, 64, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4',
                  conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'))
        last_pad = 0
        last_pad = self.model_cfg.get('last_pad', last_pad) if hasattr(self.model_cfg, 'get') else last_pad
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key='spconv_down2'),
            norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 64}

    def forward(self, batch_dict):
        return self._run(batch_dict)


class VoxelResBackBone8x(_BackboneBase):
    """spconv_backbone.py:183-293"""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        g = [int(v) for v in grid_size]
        self.sparse_shape = [g[2] + 1, g[1], g[0]]
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'),
            norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = spconv.SparseSequential(
            SparseBasicBlock(16, 16, norm_fn=norm_fn, indice_key='res1'),
            SparseBasicBlock(16, 16, norm_fn=norm_fn, indice_key='res1'))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key='res2'),
            SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key='res2'))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            SparseBasicBlock(64, 64, norm_fn=norm_fn, indice_key='res3'),
            SparseBasicBlock(64, 64, norm_fn=norm_fn, indice_key='res3'))
        self.conv4 = spconv.SparseSequential(
            block(64, 128, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4',
                  conv_type='spconv'),
            SparseBasicBlock(128, 128, norm_fn=norm_fn, indice_key='res4'),
            SparseBasicBlock(128, 128, norm_fn=norm_fn, indice_key='res4'))
        last_pad = 0
        last_pad = self.model_cfg.get('last_pad', last_pad) if hasattr(self.model_cfg, 'get') else last_pad
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(128, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key='spconv_down2'),
            norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 128}

    def forward(self, batch_dict):
        return self._run(batch_dict)
