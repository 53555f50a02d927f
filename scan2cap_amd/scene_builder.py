"""Training items assembled ON THE DEVICE from HBM-resident scenes.

Reference: `ScannetReferenceDataset.__getitem__` (lib/dataset.py:320-540) builds every item
in numpy on a DataLoader worker -- vertex sampling, channel concatenation, flips and
rotations, votes, box labels -- and the trainer copies the batch to the GPU
(lib/solver.py:280-287: 173 MB per cfg3 step, plus h5py reads of the multiview features).

Here the scenes live in device memory (`SceneStore`; the whole ScanNet train split with its
128 multiview channels is ~100 GB, a third of one MI355X) and a batch is one gather pass +
three small kernels (csrc/s2c_scene.hip).  The host keeps exactly one job: drawing the
random numbers with numpy in the reference's order (`SceneBatchBuilder.draw`), so that
an item is bit-for-bit the reference's item for the same `np.random` state.  Per step the
host sends B*N vertex indices and 32 doubles per item (2.6 MB at cfg3) in ONE pinned copy.

    store = SceneStore(device, multiview_width=128)
    store.add_scene("scene0000_00", mesh_vertices, instance_labels, semantic_labels,
                    instance_bboxes, multiview=..., rotations={object_id: 3x3})
    store.finalize()
    builder = SceneBatchBuilder(store, mean_size_arr, num_points=40000, use_normal=True,
                                use_multiview=True, augment=True)
    draws = builder.draw(scene_ids)                       # host, numpy
    data_dict = builder.build(scene_ids, object_ids, draws)   # device tensors

There is no CPU path: `build` raises `_C.S2CError` if libs2c_hip.so is missing.
Not produced: `pcl_color`, `load_time` (visualisation / logging only) and the language
entries, which `AnnotationTable` serves from resident tensors.
"""
import ctypes

import numpy as np
import torch

from . import _C

MAX_NUM_OBJ = 128        # lib/dataset.py:27
MAX_INSTANCE = 2048      # include/s2c_scene.h

# nyu40 ids with votes / boxes (model_util_scannet.py:88) and their ScanRefer class
# (model_util_scannet.py:100-115 over scannetv2-labels.combined.tsv; "others" = 17)
NYU40IDS = (3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 23, 24, 25,
            26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40)
_NAMED = {3: 0, 4: 1, 5: 2, 6: 3, 7: 4, 8: 5, 9: 6, 10: 7, 11: 8, 12: 9, 14: 10, 16: 11,
          24: 12, 28: 13, 33: 14, 34: 15, 36: 16}
CLASS_OF_NYU40 = np.full(41, -1, np.int32)
for _i in NYU40IDS:
    CLASS_OF_NYU40[_i] = _NAMED.get(_i, 17)
VOTE_ID_MASK = sum(1 << i for i in NYU40IDS)

_I, _L, _P = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p


class _Labels(ctypes.Structure):
    """s2c_scene_labels (include/s2c_scene.h)."""
    _names = ("center_label", "size_class_label", "size_residual_label", "sem_cls_label",
              "scene_object_ids", "scene_object_rotations", "scene_object_rotation_masks",
              "box_label_mask", "ref_box_label", "gt_box_corner_label", "gt_box_masks",
              "gt_box_object_ids", "num_bbox", "ref_center_label", "ref_size_class_label",
              "ref_size_residual_label", "ref_box_corner_label")
    _fields_ = [(n, _P) for n in _names]


_C.register("s2c_scene_floor_height", [_L, _P, _I, _P, _P])
_C.register("s2c_scene_sample", [_I, _I, _P, _P, _P, _P, _P])
_C.register("s2c_scene_gather", [_I] * 9 + [_P] * 9)
_C.register("s2c_scene_votes", [_I, _I, _I, _P, _P, _P, _P, _P, _P, ctypes.c_ulonglong,
                                _P, _P, _P, _P])
_C.register("s2c_scene_box_labels", [_I, _I] + [_P] * 9 + [_Labels, _P])

# (shape after (B,), dtype) of the per-box / per-item outputs
_LABEL_SPECS = {
    "center_label": ((MAX_NUM_OBJ, 3), torch.float32),
    "size_class_label": ((MAX_NUM_OBJ,), torch.int64),
    "size_residual_label": ((MAX_NUM_OBJ, 3), torch.float32),
    "sem_cls_label": ((MAX_NUM_OBJ,), torch.int64),
    "scene_object_ids": ((MAX_NUM_OBJ,), torch.int64),
    "scene_object_rotations": ((MAX_NUM_OBJ, 3, 3), torch.float32),
    "scene_object_rotation_masks": ((MAX_NUM_OBJ,), torch.int64),
    "box_label_mask": ((MAX_NUM_OBJ,), torch.float32),
    "ref_box_label": ((MAX_NUM_OBJ,), torch.int64),
    "gt_box_corner_label": ((MAX_NUM_OBJ, 8, 3), torch.float64),
    "gt_box_masks": ((MAX_NUM_OBJ,), torch.int64),
    "gt_box_object_ids": ((MAX_NUM_OBJ,), torch.int64),
    "num_bbox": ((), torch.int64),
    "ref_center_label": ((3,), torch.float32),
    "ref_size_class_label": ((), torch.int64),
    "ref_size_residual_label": ((3,), torch.float32),
    "ref_box_corner_label": ((8, 3), torch.float64),
}


def _rot(axis, t):
    """utils/pc_utils.py:282-320 (rotx / roty / rotz)."""
    c, s = np.cos(t), np.sin(t)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == "y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def _wait(event):
    """Host wait for an event by polling (the caller keeps the GIL only in slices and never
    sits inside the HIP runtime)."""
    import time
    while not event.query():
        time.sleep(0.0002)


class _SmallUploads(object):
    """Tiny host arrays -> device without a host synchronisation: a ring of pinned
    buffers (a pageable `tensor.to(device)` blocks the host until the stream gets there,
    which stalls a producer that queues work behind a running training step)."""

    def __init__(self, device, slots=8, nbytes=4096):
        self.device = torch.device(device)
        pin = self.device.type == "cuda"
        self.ring = [[torch.empty(nbytes, dtype=torch.uint8, pin_memory=pin), None]
                     for _ in range(slots)]
        self.next = 0

    def __call__(self, array):
        a = np.ascontiguousarray(array)
        rec = self.ring[self.next % len(self.ring)]
        self.next += 1
        if rec[1] is not None:
            _wait(rec[1])
        if a.nbytes > rec[0].numel():
            rec[0] = torch.empty(a.nbytes, dtype=torch.uint8,
                                 pin_memory=self.device.type == "cuda")
        rec[0].numpy()[:a.nbytes] = a.view(np.uint8).reshape(-1)
        out = rec[0][:a.nbytes].to(self.device, non_blocking=True)
        if self.device.type == "cuda":
            rec[1] = torch.cuda.Event()
            rec[1].record()
        return out.view(torch.from_numpy(a[:0]).dtype).reshape(a.shape)


class SceneStore(object):
    """Scenes concatenated in device memory.  `add_scene` takes the arrays of the
    reference's preprocessed files (data/scannet/load_scannet_data.py:147-152):
    mesh_vertices (Nv, 6|9) f32, instance / semantic labels (Nv), instance_bboxes (nb, 8)."""

    def __init__(self, device, multiview_width=0, vert_cols=9):
        self.device = torch.device(device)
        self.Cm, self.cols = int(multiview_width), int(vert_cols)
        self._host = []
        self.index = {}
        self.final = False
        self.bare = False

    def add_scene(self, scene_id, mesh_vertices, instance_labels=None, semantic_labels=None,
                  instance_bboxes=None, multiview=None, rotations=None):
        """Labels and boxes may be omitted for ALL scenes of a store (the test split,
        lib/dataset.py:611-617, loads vertices only): such a store serves
        `SceneBatchBuilder.build_clouds` but not `build`."""
        if self.final:
            raise RuntimeError("SceneStore is finalized")
        bare = instance_labels is None and semantic_labels is None and instance_bboxes is None
        if self._host and bare != self.bare:
            raise ValueError("a store holds either labelled or vertices-only scenes")
        self.bare = bare
        if bare:
            nv0 = np.asarray(mesh_vertices).shape[0]
            instance_labels = np.zeros(nv0, np.int32)
            semantic_labels = np.zeros(nv0, np.int32)
            instance_bboxes = np.array([[0, 0, 0, 1, 1, 1, 3, 0]], np.float64)   # placeholder
        v = np.ascontiguousarray(mesh_vertices, np.float32)
        nv = v.shape[0]
        if v.ndim != 2 or v.shape[1] != self.cols or nv == 0:
            raise ValueError("mesh_vertices must be (Nv>0, %d)" % self.cols)
        ins = np.asarray(instance_labels).astype(np.int64)
        sem = np.asarray(semantic_labels).astype(np.int64)
        if ins.shape != (nv,) or sem.shape != (nv,):
            raise ValueError("labels must be (Nv,)")
        if ins.min() < 0 or ins.max() >= MAX_INSTANCE:
            raise ValueError("instance ids must be in [0, %d)" % MAX_INSTANCE)
        boxes = np.ascontiguousarray(instance_bboxes, np.float64)
        if boxes.ndim != 2 or boxes.shape[1] != 8 or not 1 <= boxes.shape[0] <= MAX_NUM_OBJ:
            # the reference leaves gt_box_corner_label unbound / mis-shaped on such a
            # scene (lib/dataset.py:463-477)
            raise ValueError("a scene needs 1..%d boxes of 8 numbers" % MAX_NUM_OBJ)
        ids = boxes[:, 6].astype(np.int64)
        if ((ids < 0) | (ids > 40)).any() or (CLASS_OF_NYU40[np.clip(ids, 0, 40)] < 0).any():
            raise KeyError("box with a nyu40 id outside the 37 box classes "
                           "(DC.nyu40id2class, lib/dataset.py:445)")
        mv = None
        if self.Cm:
            mv = np.ascontiguousarray(multiview, np.float32)
            if mv.shape != (nv, self.Cm):
                raise ValueError("multiview must be (Nv, %d)" % self.Cm)
        rot = np.zeros((boxes.shape[0], 9), np.float32)
        rot_mask = np.zeros(boxes.shape[0], np.uint8)
        if rotations:
            for i, oid in enumerate(boxes[:, 7].astype(int)):   # lib/dataset.py:495-503
                r = rotations.get(int(oid), rotations.get(str(int(oid))))
                if r is not None:
                    rot[i] = np.asarray(r, np.float64).reshape(9)
                    rot_mask[i] = 1
        self.index[scene_id] = len(self._host)
        self._host.append((v, ins.astype(np.int32), sem.astype(np.int32), boxes, mv, rot,
                           rot_mask))

    def finalize(self):
        """Upload, then compute every scene's floor height on the device."""
        if not self._host:
            raise RuntimeError("SceneStore is empty")
        dev = self.device
        nvs = [h[0].shape[0] for h in self._host]
        nbs = [h[3].shape[0] for h in self._host]
        self.vert_off_host = np.concatenate([[0], np.cumsum(nvs)]).astype(np.int64)
        self.box_off_host = np.concatenate([[0], np.cumsum(nbs)]).astype(np.int32)
        self.num_vertices = np.asarray(nvs, np.int64)
        cat = lambda k: torch.from_numpy(np.concatenate([h[k] for h in self._host], 0)).to(dev)
        self.verts, self.ins, self.sem, self.boxes = cat(0), cat(1), cat(2), cat(3)
        self.mv = cat(4) if self.Cm else None
        self.box_rot, self.box_rot_mask = cat(5), cat(6)
        self.vert_off = torch.from_numpy(self.vert_off_host).to(dev)
        self.box_off = torch.from_numpy(self.box_off_host).to(dev)
        self.floor = torch.empty(len(nvs), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _C.stream_ptr()
            for s, nv in enumerate(nvs):
                _C.call("s2c_scene_floor_height", int(nv),
                        self.verts.data_ptr() + int(self.vert_off_host[s]) * self.cols * 4,
                        self.cols, self.floor.data_ptr() + 4 * s, st)
        self._host = None
        self.final = True
        return self

    def __len__(self):
        return len(self.index)

    def resident_bytes(self):
        ts = [self.verts, self.ins, self.sem, self.boxes, self.box_rot, self.box_rot_mask]
        if self.mv is not None:
            ts.append(self.mv)
        return sum(t.numel() * t.element_size() for t in ts)


class SceneBatchBuilder(object):
    """Mirror of the reference dataset's options (lib/dataset.py:288-299)."""

    def __init__(self, store, mean_size_arr, num_points=40000, use_color=False,
                 use_height=True, use_normal=False, use_multiview=False, augment=False):
        if not store.final:
            raise RuntimeError("finalize the SceneStore first")
        if use_multiview and not store.Cm:
            raise ValueError("the store holds no multiview features")
        if (use_color and store.cols < 6) or (use_normal and store.cols < 9):
            raise ValueError("the store's vertices lack the requested channels")
        self.store, self.N = store, int(num_points)
        self.use_color, self.use_height = bool(use_color), bool(use_height)
        self.use_normal, self.use_multiview = bool(use_normal), bool(use_multiview)
        self.augment = bool(augment)
        self.Cout = (3 + 3 * self.use_color + 3 * self.use_normal
                     + store.Cm * self.use_multiview + 1 * self.use_height)
        dev = store.device
        self.mean_size = torch.as_tensor(np.asarray(mean_size_arr, np.float64), device=dev)
        self.class_of = torch.from_numpy(CLASS_OF_NYU40).to(dev)
        self._staging = {}
        import os
        self.ring_slots = 3

    # ---- host: the random numbers, in the reference's order ------------------------
    def draw(self, scene_ids, rng=np.random, device_choices=False):
        """One dict per item: `choices` (utils/pc_utils.py:36-37) and, when augmenting,
        flips, the three rotation matrices and the translation (lib/dataset.py:398-424,
        :273-275).  `rng`: `np.random` (the reference's global state) or a RandomState.

        device_choices=True leaves the vertex sample to the GPU (`s2c_scene_sample`: a keyed
        Feistel permutation of the scene's vertices evaluated at 0..N-1 -- the same
        distribution: N distinct vertices in random order, with replacement only when the
        scene has fewer than N -- but not numpy's stream); the host only draws a 62-bit
        seed per item.  numpy's legacy `choice` permutes all Nv vertices per item (~2.5 ms
        for 150k on one core; the reference spreads it over DataLoader workers), which a
        12 ms training step cannot hide."""
        out = []
        for sid in scene_ids:
            nv = int(self.store.num_vertices[self.store.index[sid]])
            if device_choices:
                d = {"seed": int(rng.randint(0, 2 ** 62, dtype=np.int64))}
            else:
                d = {"choices": rng.choice(nv, self.N, replace=nv < self.N)}
            if self.augment:
                d["flip_x"] = bool(rng.random() > 0.5)
                d["flip_y"] = bool(rng.random() > 0.5)
                for ax in "xyz":
                    d["rot_" + ax] = _rot(ax, (rng.random() * np.pi / 18) - np.pi / 36)
                grid = np.arange(-0.5, 0.501, 0.001)
                d["shift"] = np.array([rng.choice(grid, size=1)[0] for _ in range(3)])
            out.append(d)
        return out

    def _pack(self, scene_ids, object_ids, draws):
        """All per-step host data in one pinned buffer: aug (B,32) f64 | object ids (B) i64
        | sampling seeds (B) u64 | scene slots (B) i32 (padded to 8 bytes) | choices (B,N)
        i64 -- the last part only when the choices were drawn on the host."""
        B, N = len(scene_ids), self.N
        o1, o2 = B * 256, B * 256 + B * 16
        o3 = o2 + ((B * 4 + 7) // 8) * 8
        host_choices = "choices" in draws[0]
        used = o3 + (B * N * 8 if host_choices else 0)
        ring = self._staging.setdefault(B, {"next": 0, "slots": []})
        if len(ring["slots"]) < self.ring_slots:
            ring["slots"].append([torch.empty(o3 + B * N * 8, dtype=torch.uint8,
                                              pin_memory=self.store.device.type == "cuda"),
                                  None])
            slot_rec = ring["slots"][-1]
        else:
            slot_rec = ring["slots"][ring["next"] % self.ring_slots]
            ring["next"] += 1
            if slot_rec[1] is not None:
                _wait(slot_rec[1])            # the H2D copy that last read this buffer
        raw = slot_rec[0].numpy()
        aug = raw[:o1].view(np.float64).reshape(B, 32)
        oid = raw[o1:o1 + B * 8].view(np.int64)
        seeds = raw[o1 + B * 8:o2].view(np.int64)
        slot = raw[o2:o2 + B * 4].view(np.int32)
        aug[:] = 0
        if host_choices:
            ch = raw[o3:used].view(np.int64).reshape(B, N)
        for b, (sid, d) in enumerate(zip(scene_ids, draws)):
            if host_choices:
                ch[b] = d["choices"]
            else:
                seeds[b] = d["seed"]
            slot[b] = self.store.index[sid]
            oid[b] = int(object_ids[b])
            if self.augment:
                aug[b, 0], aug[b, 1] = d["flip_x"], d["flip_y"]
                aug[b, 2:11] = np.asarray(d["rot_x"], np.float64).reshape(9)
                aug[b, 11:20] = np.asarray(d["rot_y"], np.float64).reshape(9)
                aug[b, 20:29] = np.asarray(d["rot_z"], np.float64).reshape(9)
                aug[b, 29:32] = d["shift"]
        return slot_rec, (o1, o2, o3, used)

    # ---- device ----------------------------------------------------------------------
    def stage(self, scene_ids, object_ids, draws):
        """Host half of `build`: packs the draws and starts their (pinned, asynchronous)
        copy to the device on the current stream.  Touches no output buffer, so a producer
        can issue it before it waits for its output buffers to become free."""
        B = len(scene_ids)
        if not (len(object_ids) == len(draws) == B) or B == 0:
            raise ValueError("scene_ids, object_ids and draws must have one entry per item")
        rec, offs = self._pack(scene_ids, object_ids, draws)
        with torch.cuda.device(self.store.device):
            d = rec[0][:offs[3]].to(self.store.device, non_blocking=True)
            rec[1] = torch.cuda.Event()
            rec[1].record()
        return B, d, offs

    def build(self, scene_ids, object_ids, draws, out=None, staged=None):
        """-> data_dict of device tensors with the reference's keys and dtypes.
        `out`: a data_dict whose tensors are written in place where the key exists (static
        buffers of a captured hipGraph); a shape / dtype mismatch raises.
        `staged`: the result of `stage(...)` for the same arguments."""
        st = self.store
        dev = st.device
        if st.bare:
            raise ValueError("the store holds vertices only: use build_clouds")
        if staged is None:
            staged = self.stage(scene_ids, object_ids, draws)
        B, d, (o1, o2, o3, used) = staged
        N = self.N
        given = out if out is not None else {}

        def buf(key, shape, dtype, zero=False):
            t = given.get(key)
            if t is None:
                return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=dev)
            if tuple(t.shape) != tuple(shape) or t.dtype != dtype or not t.is_contiguous():
                raise ValueError("out[%r] must be a contiguous %s tensor of shape %s"
                                 % (key, dtype, tuple(shape)))
            return t.zero_() if zero else t
        with torch.cuda.device(dev):
            aug = d[:o1].view(torch.float64)
            oid = d[o1:o1 + B * 8].view(torch.int64)
            seeds = d[o1 + B * 8:o2].view(torch.int64)
            slot = d[o2:o2 + B * 4].view(torch.int32)
            s = _C.stream_ptr()
            if used > o3:
                choices = d[o3:used].view(torch.int64).view(B, N)
            else:
                choices = torch.empty((B, N), dtype=torch.int64, device=dev)
                _C.call("s2c_scene_sample", B, N, st.vert_off.data_ptr(), slot.data_ptr(),
                        seeds.data_ptr(), choices.data_ptr(), s)
            cloud = buf("point_clouds", (B, N, self.Cout), torch.float32)
            _C.TIMER.alg_bytes = 8 * B * N * self.Cout + 8 * B * N
            _C.call("s2c_scene_gather", B, N, st.cols, st.Cm, int(self.use_color),
                    int(self.use_normal), int(self.use_multiview), int(self.use_height),
                    int(self.augment), st.verts.data_ptr(),
                    st.mv.data_ptr() if st.mv is not None else None, st.vert_off.data_ptr(),
                    st.floor.data_ptr(), slot.data_ptr(), choices.data_ptr(),
                    aug.data_ptr(), cloud.data_ptr(), s)
            votes = buf("vote_label", (B, N, 9), torch.float32)
            vmask = buf("vote_label_mask", (B, N), torch.int64)
            work = torch.empty(B * MAX_INSTANCE * 7, dtype=torch.int32, device=dev)
            _C.call("s2c_scene_votes", B, N, self.Cout, cloud.data_ptr(), st.ins.data_ptr(),
                    st.sem.data_ptr(), st.vert_off.data_ptr(), slot.data_ptr(),
                    choices.data_ptr(), VOTE_ID_MASK, work.data_ptr(), votes.data_ptr(),
                    vmask.data_ptr(), s)
            res = {k: buf(k, (B,) + shp, dt) for k, (shp, dt) in _LABEL_SPECS.items()}
            lab = _Labels(**{k: res[k].data_ptr() for k in _Labels._names})
            _C.call("s2c_scene_box_labels", B, int(self.augment), st.boxes.data_ptr(),
                    st.box_off.data_ptr(), st.box_rot.data_ptr(), st.box_rot_mask.data_ptr(),
                    slot.data_ptr(), oid.data_ptr(), aug.data_ptr(), self.class_of.data_ptr(),
                    self.mean_size.data_ptr(), lab, s)
            res.update(
                point_clouds=cloud, vote_label=votes, vote_label_mask=vmask,
                heading_class_label=buf("heading_class_label", (B, MAX_NUM_OBJ), torch.int64, True),
                heading_residual_label=buf("heading_residual_label", (B, MAX_NUM_OBJ),
                                           torch.float32, True),
                ref_heading_class_label=buf("ref_heading_class_label", (B,), torch.int64, True),
                ref_heading_residual_label=buf("ref_heading_residual_label", (B,), torch.int64,
                                               True))
            res["object_id"] = buf("object_id", (B,), torch.int64).copy_(oid)
            res["_choices"] = choices          # the sampled vertex of every cloud row
        return res

    def build_clouds(self, scene_ids, draws, out=None):
        """`point_clouds` only -- the item of the reference's TEST dataset
        (`ScannetReferenceTestDataset.__getitem__`, lib/dataset.py:567-609: sampling and the
        feature channels, no augmentation, no labels).  -> {"point_clouds": (B,N,3+C)}"""
        st = self.store
        dev = st.device
        B, d, (o1, o2, o3, used) = self.stage(scene_ids, [0] * len(scene_ids), draws)
        N = self.N
        with torch.cuda.device(dev):
            slot = d[o2:o2 + B * 4].view(torch.int32)
            seeds = d[o1 + B * 8:o2].view(torch.int64)
            s = _C.stream_ptr()
            if used > o3:
                choices = d[o3:used].view(torch.int64).view(B, N)
            else:
                choices = torch.empty((B, N), dtype=torch.int64, device=dev)
                _C.call("s2c_scene_sample", B, N, st.vert_off.data_ptr(), slot.data_ptr(),
                        seeds.data_ptr(), choices.data_ptr(), s)
            cloud = out["point_clouds"] if out is not None and "point_clouds" in out else \
                torch.empty((B, N, self.Cout), dtype=torch.float32, device=dev)
            if tuple(cloud.shape) != (B, N, self.Cout) or cloud.dtype != torch.float32 \
                    or not cloud.is_contiguous():
                raise ValueError("out['point_clouds'] must be a contiguous float32 (B,N,%d) tensor"
                                 % self.Cout)
            _C.TIMER.alg_bytes = 8 * B * N * self.Cout + 8 * B * N
            _C.call("s2c_scene_gather", B, N, st.cols, st.Cm, int(self.use_color),
                    int(self.use_normal), int(self.use_multiview), int(self.use_height), 0,
                    st.verts.data_ptr(), st.mv.data_ptr() if st.mv is not None else None,
                    st.vert_off.data_ptr(), st.floor.data_ptr(), slot.data_ptr(),
                    choices.data_ptr(), d[:o1].view(torch.float64).data_ptr(), cloud.data_ptr(), s)
        return {"point_clouds": cloud, "_choices": choices}


class AnnotationTable(object):
    """The language side of an item (lib/dataset.py:326-329, :505-508, :531-534) served
    from resident tensors: `lang_feat` (A,32,300) f32, `lang_ids` (A,32) i64, `lang_len`
    (A) i64 (already clipped to MAX_DES_LEN + 2), plus per-annotation integers."""

    def __init__(self, device, lang_feat, lang_ids, lang_len, object_id, ann_id,
                 object_cat, unique_multiple):
        dev = torch.device(device)
        t = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt).to(dev)
        self.t = {"lang_feat": t(lang_feat, torch.float32), "lang_ids": t(lang_ids, torch.int64),
                  "lang_len": t(lang_len, torch.int64), "object_id": t(object_id, torch.int64),
                  "ann_id": t(ann_id, torch.int64), "object_cat": t(object_cat, torch.int64),
                  "unique_multiple": t(unique_multiple, torch.int64)}
        self.object_id_host = np.asarray(object_id, np.int64)
        self.device = dev
        self._upload = _SmallUploads(dev)

    def gather(self, indices):
        idx = self._upload(np.asarray(indices, np.int64))
        out = {k: v.index_select(0, idx) for k, v in self.t.items()}
        out["dataset_idx"] = idx
        return out


class BatchFeeder(object):
    """Producer for a captured training step: batches are assembled into static buffer sets
    (`buffers[p]` = the data_dict the p-th graph was captured on; only keys present in it
    are written) on the feeder's own stream while other steps run.

        feeder.acquire(p); graphs[p].replay(); feeder.release(p)
        feeder.produce(q, scene_ids, object_ids, ann_indices, host_wait=True)

    Measured in `bench.py --feed builder` (cfg3 train step, 12.5 ms with one resident
    batch): with TWO sets and the producer's stream parked on the event that frees its set
    (plus the geometry stream parked on the producer) the step takes 14.2 ms although
    neither is on the critical path -- every stream parked on a cross-stream event cost the
    training stream 0.3-0.7 ms per step on this stack.  With THREE sets and host_wait=True
    (the host itself waits for step i-1 before enqueuing batch i+2, so nothing ever
    parks) the same work costs 12.9 ms."""

    def __init__(self, builder, annotations, buffers, device_choices=True, rng=None,
                 stream=None):
        """stream: the producer's stream; pass one that shares no hardware queue with the
        training stream (pipeline.independent_streams) -- the producer parks on an event
        until the previous reader of a buffer set is done, and a parked stream holds up
        every other stream mapped to the same queue."""
        self.builder, self.annotations, self.buffers = builder, annotations, buffers
        self.device_choices = device_choices
        self.rng = rng if rng is not None else np.random.RandomState(0)
        self.stream = stream if stream is not None else torch.cuda.Stream(builder.store.device)
        self.ready = [None] * len(buffers)
        self.consumed = [None] * len(buffers)

    def produce(self, p, scene_ids, object_ids, ann_indices=None, host_wait=False):
        """host_wait=True: the HOST waits (polling) until the last reader of set p is done and
        only then enqueues the work, so no stream ever parks on an event (with three buffer
        sets the host then runs one step ahead of the GPU)."""
        with torch.cuda.stream(self.stream):
            draws = self.builder.draw(scene_ids, rng=self.rng,
                                      device_choices=self.device_choices)
            staged = self.builder.stage(scene_ids, object_ids, draws)
            if self.consumed[p] is not None:
                if host_wait:
                    _wait(self.consumed[p])
                self.stream.wait_event(self.consumed[p])      # the reader of set p is done
            self.builder.build(scene_ids, object_ids, draws, out=self.buffers[p],
                               staged=staged)
            if ann_indices is not None and self.annotations is not None:
                for k, v in self.annotations.gather(ann_indices).items():
                    t = self.buffers[p].get(k)
                    if torch.is_tensor(t):
                        t.copy_(v)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.ready[p] = ev
        return ev

    def acquire(self, p):
        torch.cuda.current_stream().wait_event(self.ready[p])

    def release(self, p):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.consumed[p] = ev
