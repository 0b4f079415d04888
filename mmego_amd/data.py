"""Sample_data loader: per-frame .mat files -> fixed-size windows of radar points, skeletons, IMU and head pose.

Counterpart of the reference's Util/Universal_Util/Dataset_sample.py:12-277 (class PosePC) -- same constructor,
same per-item tuples, same array attributes.  The numpy global RNG is consumed in the same order as the
reference (one choice() per frame for the slot layout, plus the reference's unused second draw), so
``np.random.seed(k)`` before construction reproduces the reference's padded point clouds bit for bit
(checked in tests/test_data_cpu.py against tests/golden/real16.npz).
"""
import glob
import hashlib
import os
import re

import numpy as np
import scipy.io as scio

from .config import Config

CACHE_VERSION = 1

R_RI = np.array([[0, 0, 1], [0, -1, 0], [1, 0, 0]])
R_TTB = np.array([[0, -1, 0], [-1, 0, 0], [0, 0, -1]])
R_CTW = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]])


def _numeric_key(path):
    return [int(tok) for tok in re.findall(r"\d+", os.path.basename(path))]


def list_snippets(root):
    """[(action_index, snippet_index, [frame files...])] in the reference's traversal order."""
    out = []
    actions = sorted(os.listdir(root), key=lambda name: int(name))
    for a, action in enumerate(actions):
        adir = os.path.join(root, action)
        for j, snip in enumerate(sorted(os.listdir(adir))):
            sdir = os.path.join(adir, snip)
            if not os.path.isdir(sdir):
                continue
            frames = sorted(glob.glob(os.path.join(sdir, "*.mat")), key=_numeric_key)
            if frames and not (a == 0 and j == 0):           # the reference skips the very first snippet
                out.append((a, j, frames))
    return out


def pack_points(raw, pc_no):
    """(n,5) x,y,z,intensity,velocity -> (pc_no,6) x,y,z,range,velocity,intensity, zero-padded at random slots
    or randomly subsampled (Dataset_sample.py:203-224)."""
    n = raw.shape[0]
    pts = np.zeros((n, 6), dtype=np.float32)
    pts[:, 0:3] = raw[:, :3]
    pts[:, 3] = np.linalg.norm(raw[:, 0:3], axis=1)
    pts[:, 4] = raw[:, 4]
    pts[:, 5] = raw[:, 3]
    if n < pc_no:
        slots = np.random.choice(pc_no, size=n, replace=False)
        np.random.choice(n, size=pc_no - n, replace=True)      # drawn and discarded by the reference; keeps RNG in step
        frame = np.zeros((pc_no, 6), dtype=np.float32)
        frame[slots] = pts
        return frame
    return pts[np.random.choice(n, size=pc_no, replace=False)]


def imu_to_radar_frame(imu, orientation_ref):
    """Rotate the 20x(9+3+3) IMU samples into the radar frame and fix signs/gravity (Dataset_sample.py:183-192).
    Mutates and returns ``imu``."""
    R_ni = np.stack([imu[:, :3], imu[:, 3:6], imu[:, 6:9]], axis=2)
    rot = R_RI @ (orientation_ref.T @ R_ni) @ R_RI.T
    imu[:, :3], imu[:, 3:6], imu[:, 6:9] = rot[:, 0, :], rot[:, 1, :], rot[:, 2, :]
    imu[:, 11] = imu[:, 11] + 9.8
    imu[:, 10:12] = -1 * imu[:, 10:12]
    imu[:, 13:] = -1 * imu[:, 13:]
    return imu


class PosePC:
    """Dataset of non-overlapping ``frame_no``-frame windows (taken from the tail of each recording)."""

    def __init__(self, train=True, vis=False, batch_length=None, root=None):
        self.vis, self.train = vis, train
        self.pc_no = Config.pc_no
        self.frame_no = batch_length if batch_length is not None else Config.frame_no
        self.joint_selection = Config.kinect_joint_selection
        self.skeleton = Config.skeleton_all.tolist()
        self.root = root or Config.data_root
        (self.data_ti_, self.data_key_, self.imu_, self.skl_, self.ground_, self.foot_contact_, self.R_R0R_,
         self.t_R0R_, self.R_RtW_) = self._read()
        n = len(self.data_ti_)
        if not vis:
            order = np.arange(n)
            np.random.RandomState(Config.dataset_random_seed).shuffle(order)    # same permutation as shuffling each array
            for name in ("data_ti_", "data_key_", "skl_", "ground_", "foot_contact_", "imu_", "R_R0R_", "t_R0R_"):
                setattr(self, name, getattr(self, name)[order])
        cut = int(n * 0.8)
        sl = slice(0, n) if vis else (slice(0, cut) if train else slice(cut, n))
        self._items = [getattr(self, k)[sl] for k in ("data_ti_", "data_key_", "skl_", "imu_", "ground_",
                                                      "foot_contact_", "R_R0R_", "t_R0R_")]
        if vis:
            self._items.append(self.R_RtW_)

    def __len__(self):
        return len(self._items[0])

    def __getitem__(self, i):
        return tuple(a[i] for a in self._items)

    # -----------------------------------------------------------------------------------------
    def _decode(self):
        """Everything that comes out of the .mat files and does NOT depend on numpy's RNG: per snippet, per non-empty
        frame the raw points and the derived small arrays.  This is the slow part of the reference's start-up (one
        scipy.loadmat per frame, 19 208 files = 14 s, done twice: train and test set) and it is deterministic, so it is
        kept in a packed binary cache: one .npz per (file list, sizes, mtimes)."""
        snippets = list_snippets(self.root)
        cache = _cache_path(self.root, snippets, self.joint_selection, self.skeleton)
        if cache and os.path.exists(cache):
            try:
                z = np.load(cache, allow_pickle=False)
                return {k: z[k] for k in z.files}
            except (OSError, ValueError, KeyError):
                pass
        fr = {k: [] for k in ("pts", "npts", "key", "imu", "ground", "foot", "R", "t", "RtW")}
        snip_len = []
        ref = None          # (R_btc, imu orientation, bone vectors) of the first frame ever read
        for _, _, files in snippets:
            count = 0
            for path in files:
                m = scio.loadmat(path)
                raw = np.asarray(m["pc_xyziv_ti2"][:, 0:5].tolist())
                if len(raw) == 0:
                    continue
                joints = np.asarray([m["pc_xyz_key_2"][:, 0:3][i] for i in self.joint_selection])
                imu = m["imu_save_l"]
                if ref is None:
                    bones = [joints[p] - joints[c] for p, c in self.skeleton]
                    ref = (m["R_btc"], np.asarray(m["orientation_imu_img"]), bones)
                R_btc = m["R_btc"]
                fr["R"].append(R_TTB @ ref[0] @ R_btc.T @ R_TTB.T)
                fr["RtW"].append(R_TTB @ R_btc @ R_CTW)
                fr["imu"].append(imu_to_radar_frame(imu, ref[1]))
                fc = m["foot_contact"]
                fr["foot"].append([[0, 1] if fc[0, 0] else [1, 0], [0, 1] if fc[0, 1] else [1, 0]])
                ground = m["abcd_ground_2"]
                fr["ground"].append(-1 * ground if ground[0, 0] > 0 else ground)
                fr["pts"].append(raw)
                fr["npts"].append(len(raw))
                fr["key"].append(joints)
                fr["t"].append(m["t_R0R"])
                count += 1
            snip_len.append(count)
        dec = {"pts": np.concatenate(fr["pts"], 0) if fr["pts"] else np.zeros((0, 5)), "npts": np.asarray(fr["npts"], dtype=np.int64),
               "snip_len": np.asarray(snip_len, dtype=np.int64),
               "bones": np.asarray(ref[2]) if ref is not None else np.zeros((0, 3))}
        for k in ("key", "imu", "ground", "foot", "R", "t", "RtW"):
            dec[k] = np.asarray(fr[k])
        if cache:
            try:
                os.makedirs(os.path.dirname(cache), exist_ok=True)
                tmp = cache + ".tmp%d.npz" % os.getpid()
                np.savez(tmp, **dec)
                os.replace(tmp, cache)
            except OSError:
                pass                                    # read-only location: work without a cache
        return dec

    def _read(self):
        if not os.path.isdir(self.root):
            raise FileNotFoundError("Sample_data not found at %s (set MMEGO_DATA_ROOT or Config.data_root)" % self.root)
        dec = self._decode()
        win = {k: [] for k in ("ti", "key", "imu", "skl", "ground", "foot", "R", "t", "RtW")}
        bones = [b for b in dec["bones"]]
        f0 = 0                                           # first frame of the snippet in the packed arrays
        p0 = 0                                           # first point of that frame in dec["pts"]
        for count in dec["snip_len"]:
            rec = {k: [] for k in ("ti", "key", "imu", "ground", "foot", "R", "t", "RtW")}
            for f in range(f0, f0 + int(count)):
                n = int(dec["npts"][f])
                rec["ti"].append(pack_points(dec["pts"][p0:p0 + n], self.pc_no))     # consumes numpy's RNG like the reference
                p0 += n
                for k in ("key", "imu", "ground", "foot", "R", "t", "RtW"):
                    rec[k].append(dec[k][f])
            f0 += int(count)
            L = self.frame_no
            while len(rec["ti"]) >= L:
                for k in rec:
                    win[k].append(rec[k][-L:])
                    rec[k] = rec[k][:-L]
                win["skl"].append(bones)
        print("data load end")
        return tuple(np.asarray(win[k]) for k in ("ti", "key", "imu", "skl", "ground", "foot", "R", "t", "RtW"))


def _cache_path(root, snippets, joint_selection, skeleton):
    """Cache file of the decoded frames of ``root``: keyed by every frame file's path, size and mtime (and by the joint
    selection).  MMEGO_CACHE_DIR overrides the location (default ~/.cache/mmego_amd); MMEGO_CACHE_DIR=off disables it."""
    base = os.environ.get("MMEGO_CACHE_DIR", os.path.join(os.path.expanduser("~"), ".cache", "mmego_amd"))
    if base == "off":
        return None
    h = hashlib.sha256()
    h.update(("v%d|%s|%s|%s" % (CACHE_VERSION, os.path.abspath(root), list(joint_selection), skeleton)).encode())
    for _, _, files in snippets:
        for path in files:
            st = os.stat(path)
            h.update(("%s|%d|%d" % (path, st.st_size, st.st_mtime_ns)).encode())
    return os.path.join(base, "frames_%s.npz" % h.hexdigest()[:24])


class DeviceArrays:
    """The training arrays of a PosePC resident in HBM as fp32 (uploaded once: 835 x 20 x 128 x 6 floats = 51 MB for
    Sample_data), with minibatches gathered ON the device by index (mmego_gather_rows).  Replaces the reference's
    per-batch numpy stacking + float64->float32 `torch.tensor(...)` + host->device copies (Train_Upper.py:140-150);
    the values are the same (the one rounding to fp32 happens at upload instead of per batch)."""

    FIELDS = (("data", 0), ("target", 1), ("skl", 2), ("imu", 3), ("R_R0R", 6))

    def __init__(self, dataset, device):
        import torch
        self.n = len(dataset)
        self.device = device
        self.src, self.shape = {}, {}
        for name, i in self.FIELDS:
            a = np.ascontiguousarray(dataset._items[i])
            self.shape[name] = tuple(a.shape[1:])
            self.src[name] = torch.as_tensor(a.reshape(self.n, -1), dtype=torch.float32).to(device)
        self._out = {}

    @staticmethod
    def of(dataset, device):
        """The device copy of ``dataset``, uploaded once per dataset OBJECT and shared by everybody who evaluates or trains on
        it.  The copy hangs on the dataset itself (not in a table keyed by id(): an id can be reused after garbage collection), is
        dropped with it, and is rebuilt when the dataset's length or its arrays' identity changed (a split mutated in place)."""
        stamp = (str(device), len(dataset)) + tuple(id(dataset._items[i]) for _, i in DeviceArrays.FIELDS)
        ent = dataset.__dict__.get("_device_arrays")
        if ent is None or ent[0] != stamp:
            ent = dataset.__dict__["_device_arrays"] = (stamp, DeviceArrays(dataset, device))
        return ent[1]

    def gather(self, index):
        """index: int array of item numbers -> dict of device tensors [len(index), ...] (buffers reused per batch size)."""
        import torch
        from . import ops
        idx = torch.as_tensor(np.ascontiguousarray(index), dtype=torch.int64).to(self.device)
        B = idx.numel()
        out = {}
        for name, _ in self.FIELDS:
            key = (name, B)
            dst = self._out.get(key)
            if dst is None:
                dst = torch.empty((B, self.src[name].shape[1]), dtype=torch.float32, device=self.device)
                self._out[key] = dst
            ops.gather_rows(self.src[name], idx, dst)
            out[name] = dst.view((B,) + self.shape[name])
        return out


def _gather_field_into(self, name, index, out):
    """One field of the items `index` into a caller-owned buffer `out` [len(index), ...] (the prefetch of the NEXT minibatch's
    IMU samples for train_step.PipelinedStages, which must not disturb the current minibatch's buffers)."""
    import torch
    from . import ops
    idx = torch.as_tensor(np.ascontiguousarray(index), dtype=torch.int64).to(self.device)
    ops.gather_rows(self.src[name], idx, out.view(idx.numel(), -1))
    return out


DeviceArrays.gather_field_into = _gather_field_into


class ArraySplit:
    """A data split given as arrays (what PosePC exposes to the trainers: `_items` in the reference loader's tuple order
    data, target, skl, imu, ground, foot_contact, R_R0R, t_R0R -- Dataset_sample.py:264-277).  For callers that hold windows
    already (tests, fixtures)."""

    def __init__(self, data, target, skl, imu, R_R0R):
        n = len(data)
        z = np.zeros((n, 1), dtype=np.float32)
        self._items = [np.asarray(data), np.asarray(target), np.asarray(skl), np.asarray(imu), z, z, np.asarray(R_R0R), z]

    def __len__(self):
        return len(self._items[0])

    def __getitem__(self, i):
        return tuple(a[i] for a in self._items)


def batch_indices(n, batch_size, shuffle, rng=None):
    """The index sets `batches` iterates over (same RNG consumption), for on-device gathering."""
    order = (rng or np.random).permutation(n) if shuffle else np.arange(n)
    for s in range(0, n, batch_size):
        yield order[s:s + batch_size]


def batches(dataset, batch_size, shuffle, rng=None):
    """Minibatch iterator (the reference uses torch DataLoader(num_workers=0, drop_last=False)): yields tuples of
    stacked numpy arrays."""
    n = len(dataset)
    order = (rng or np.random).permutation(n) if shuffle else np.arange(n)
    for s in range(0, n, batch_size):
        idx = order[s:s + batch_size]
        yield tuple(a[idx] for a in dataset._items)
