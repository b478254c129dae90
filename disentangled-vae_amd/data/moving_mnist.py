"""Synthetic Moving-MNIST clips: seeded trajectories of MNIST digits bouncing in a square canvas.

Bit-exact with the reference generator (data/moving_mnist.py:84-221) for the same digit bank: same
RandomState seeding (seed + index, with seed 0 treated as "no seed"), same draw order, float64
trajectory state, truncation to integer pixel positions and max-compositing.

The digit bank is [N,28,28] float32 = uint8/255.  It comes from torchvision's MNIST when that package
is importable, from the raw IDX files under <root>/MNIST/raw otherwise, or from `digit_bank=` (an
array passed in by the caller, used by the tests and the benchmark where MNIST is unavailable).
"""
import gzip
import os
import struct
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset


def _read_idx(path: str) -> np.ndarray:
    opener = gzip.open if path.endswith('.gz') else open
    with opener(path, 'rb') as f:
        data = f.read()
    magic, = struct.unpack('>I', data[:4])
    ndim = magic & 0xFF
    dims = struct.unpack('>' + 'I' * ndim, data[4:4 + 4 * ndim])
    return np.frombuffer(data, dtype=np.uint8, offset=4 + 4 * ndim).reshape(dims)


def load_mnist_bank(root: str, train: bool, download: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """(images [N,28,28] float32 in [0,1], labels [N] int64)."""
    try:
        from torchvision import datasets                        # noqa: WPS433 (optional dependency)
        ds = datasets.MNIST(root=root, train=train, download=download)
        return ds.data.numpy().astype(np.float32) / 255.0, ds.targets.numpy().astype(np.int64)
    except ImportError:
        pass
    stem = 'train' if train else 't10k'
    raw = os.path.join(root, 'MNIST', 'raw')
    for ext in ('', '.gz'):
        pi, pl = (os.path.join(raw, f'{stem}-images-idx3-ubyte{ext}'),
                  os.path.join(raw, f'{stem}-labels-idx1-ubyte{ext}'))
        if os.path.exists(pi) and os.path.exists(pl):
            return _read_idx(pi).astype(np.float32) / 255.0, _read_idx(pl).astype(np.int64)
    raise FileNotFoundError(
        f"MNIST not found: torchvision is not installed and {raw} holds no IDX files. Put the four MNIST "
        "IDX files there, or pass digit_bank= (and digit_labels=) explicitly.")


class MovingMNIST(Dataset):

    def __init__(self, root: str = './data', train: bool = True, num_videos: int = 10000, sequence_length: int = 20,
                 frame_size: int = 64, num_digits: int = 2, digit_size: int = 28,
                 velocity_range: Tuple[float, float] = (2.0, 5.0), transform=None, seed: Optional[int] = None,
                 deterministic: bool = False, download: bool = True, digit_bank: Optional[np.ndarray] = None,
                 digit_labels: Optional[np.ndarray] = None):
        super().__init__()
        self.num_videos = num_videos
        self.sequence_length = sequence_length
        self.frame_size = frame_size
        self.num_digits = num_digits
        self.digit_size = digit_size
        self.velocity_range = velocity_range
        self.transform = transform
        self.deterministic = deterministic
        self.seed = seed
        if digit_bank is not None:
            self.digit_images = np.ascontiguousarray(digit_bank, dtype=np.float32)
            self.digit_labels = (np.asarray(digit_labels, dtype=np.int64) if digit_labels is not None
                                 else np.arange(len(self.digit_images), dtype=np.int64) % 10)
        else:
            self.digit_images, self.digit_labels = load_mnist_bank(root, train, download)
        self.label_to_indices = {i: np.where(self.digit_labels == i)[0] for i in range(10)}
        self.rng = np.random.RandomState(seed) if seed is not None else np.random.RandomState()
        self._batch_rng = np.random.RandomState(0)      # re-seeded per clip by simulate_batch

    def __len__(self) -> int:
        return self.num_videos

    # ---- draws, in the reference's order ----------------------------------------------------
    def _get_random_digit(self, rng, label: Optional[int] = None):
        idx = rng.choice(self.label_to_indices[label]) if label is not None else rng.randint(len(self.digit_images))
        return self.digit_images[idx], self.digit_labels[idx]

    def _get_random_position(self, rng) -> np.ndarray:
        hi = self.frame_size - self.digit_size
        x = rng.uniform(0, hi)
        y = rng.uniform(0, hi)
        return np.array([x, y])

    def _get_random_velocity(self, rng) -> np.ndarray:
        speed = rng.uniform(self.velocity_range[0], self.velocity_range[1])
        angle = rng.uniform(0, 2 * np.pi)
        return np.array([speed * np.cos(angle), speed * np.sin(angle)])

    def _bounce(self, position: np.ndarray, velocity: np.ndarray):
        hi = self.frame_size - self.digit_size
        p = position + velocity
        v = velocity.copy()
        for axis in (0, 1):
            if p[axis] < 0:
                p[axis] = -p[axis]
                v[axis] = -v[axis]
            elif p[axis] > hi:
                p[axis] = 2 * hi - p[axis]
                v[axis] = -v[axis]
        return np.clip(p, 0, hi), v

    def _render_frame(self, digit_images: List[np.ndarray], positions: List[np.ndarray]) -> np.ndarray:
        S, D = self.frame_size, self.digit_size
        frame = np.zeros((S, S), dtype=np.float32)
        for img, pos in zip(digit_images, positions):
            x, y = int(pos[0]), int(pos[1])
            x0, y0, x1, y1 = max(0, x), max(0, y), min(S, x + D), min(S, y + D)
            patch = img[y0 - y:y0 - y + (y1 - y0), x0 - x:x0 - x + (x1 - x0)]
            np.maximum(frame[y0:y1, x0:x1], patch, out=frame[y0:y1, x0:x1])
        return frame

    def simulate(self, rng, with_meta: bool = True):
        """The draws and the bouncing trajectories of one clip without rendering it: (bank indices
        [num_digits], integer top-left positions [T, num_digits, 2] as the renderer truncates them,
        metadata dict identical to generate_video's, or None).  Same draw order and the same float64
        arithmetic as generate_video, on Python scalars instead of 2-element arrays."""
        nd, T = self.num_digits, self.sequence_length
        hi = self.frame_size - self.digit_size
        idxs = [rng.randint(len(self.digit_images)) for _ in range(nd)]
        pos = []
        for _ in range(nd):
            x = rng.uniform(0, hi)
            y = rng.uniform(0, hi)
            pos.append([float(x), float(y)])
        vel = []
        for _ in range(nd):
            speed = rng.uniform(self.velocity_range[0], self.velocity_range[1])
            angle = rng.uniform(0, 2 * np.pi)
            vel.append([float(speed * np.cos(angle)), float(speed * np.sin(angle))])
        ipos = np.empty((T, nd, 2), dtype=np.int32)
        hist = np.empty((2, T, nd, 2), dtype=np.float64) if with_meta else None
        for t in range(T):
            for i in range(nd):
                p, v = pos[i], vel[i]
                ipos[t, i, 0], ipos[t, i, 1] = int(p[0]), int(p[1])
                if with_meta:
                    hist[0, t, i, 0], hist[0, t, i, 1] = p[0], p[1]
                    hist[1, t, i, 0], hist[1, t, i, 1] = v[0], v[1]
                for axis in (0, 1):
                    q = p[axis] + v[axis]
                    if q < 0:
                        q = -q
                        v[axis] = -v[axis]
                    elif q > hi:
                        q = 2 * hi - q
                        v[axis] = -v[axis]
                    p[axis] = min(max(q, 0.0), float(hi))                 # np.clip(q, 0, hi)
        meta = None
        if with_meta:
            pos_hist = [[hist[0, t, i].copy() for i in range(nd)] for t in range(T)]
            vel_hist = [[hist[1, t, i].copy() for i in range(nd)] for t in range(T)]
            meta = {'labels': [self.digit_labels[i] for i in idxs],
                    'initial_positions': [p.copy() for p in pos_hist[0]],
                    'initial_velocities': [v.copy() for v in vel_hist[0]],
                    'position_history': pos_hist, 'velocity_history': vel_hist}
        return np.asarray(idxs, dtype=np.int32), ipos, meta

    def simulate_batch(self, indices, with_meta: bool = True):
        """simulate() for the clips `indices` of a deterministic dataset at once: the seeded draws are
        made per clip (one re-seeded generator, three vector draws that consume the stream exactly like
        the scalar calls of generate_video), the T bounce steps run vectorised over the batch in the
        same float64 arithmetic.  Returns (bank indices [B, num_digits] int32, positions
        [B, T, num_digits, 2] int32, list of metadata dicts or empty dicts)."""
        nd, T, B = self.num_digits, self.sequence_length, len(indices)
        hi = self.frame_size - self.digit_size
        lo_v, hi_v = self.velocity_range
        didx = np.empty((B, nd), dtype=np.int64)
        pos = np.empty((B, nd, 2), dtype=np.float64)
        sa = np.empty((B, nd, 2), dtype=np.float64)                  # (speed, angle) draws
        rs = self._batch_rng
        n_bank = len(self.digit_images)
        for j, i in enumerate(indices):
            rs.seed(self.seed + i if self.seed else i)
            didx[j] = rs.randint(n_bank, size=nd)
            pos[j] = rs.uniform(0, hi, size=2 * nd).reshape(nd, 2)
            sa[j] = rs.random_sample(2 * nd).reshape(nd, 2)
        speed = lo_v + (hi_v - lo_v) * sa[..., 0]                  # uniform(low, high) = low + (high-low)*u
        angle = 0 + (2 * np.pi - 0) * sa[..., 1]
        vel = np.stack([speed * np.cos(angle), speed * np.sin(angle)], axis=-1)
        ipos = np.empty((B, T, nd, 2), dtype=np.int32)
        hist = np.empty((2, T, B, nd, 2), dtype=np.float64) if with_meta else None
        for t in range(T):
            ipos[:, t] = pos.astype(np.int32)                        # int(): truncation, positions are >= 0
            if with_meta:
                hist[0, t], hist[1, t] = pos, vel
            p = pos + vel
            neg = p < 0
            over = (p > hi) & ~neg
            p = np.where(neg, -p, np.where(over, 2 * hi - p, p))
            vel = np.where(neg | over, -vel, vel)
            pos = np.clip(p, 0, hi)
        metas = [{} for _ in range(B)]
        if with_meta:
            for j in range(B):
                ph = [[hist[0, t, j, k].copy() for k in range(nd)] for t in range(T)]
                vh = [[hist[1, t, j, k].copy() for k in range(nd)] for t in range(T)]
                metas[j] = {'labels': [self.digit_labels[k] for k in didx[j]],
                            'initial_positions': [x.copy() for x in ph[0]],
                            'initial_velocities': [x.copy() for x in vh[0]],
                            'position_history': ph, 'velocity_history': vh}
        return didx.astype(np.int32), ipos, metas

    def generate_video(self, rng, digit_labels: Optional[List[int]] = None,
                       initial_positions: Optional[List[np.ndarray]] = None,
                       initial_velocities: Optional[List[np.ndarray]] = None):
        imgs, labels = [], []
        for i in range(self.num_digits):
            img, lbl = self._get_random_digit(rng, digit_labels[i] if digit_labels else None)
            imgs.append(img)
            labels.append(lbl)
        pos = ([p.copy() for p in initial_positions] if initial_positions is not None
               else [self._get_random_position(rng) for _ in range(self.num_digits)])
        vel = ([v.copy() for v in initial_velocities] if initial_velocities is not None
               else [self._get_random_velocity(rng) for _ in range(self.num_digits)])
        T = self.sequence_length
        video = np.empty((T, 1, self.frame_size, self.frame_size), dtype=np.float32)
        pos_hist, vel_hist = [], []
        for t in range(T):
            video[t, 0] = self._render_frame(imgs, pos)
            pos_hist.append([p.copy() for p in pos])
            vel_hist.append([v.copy() for v in vel])
            for i in range(self.num_digits):
                pos[i], vel[i] = self._bounce(pos[i], vel[i])
        meta = {'labels': labels,
                'initial_positions': [p.copy() for p in pos_hist[0]],
                'initial_velocities': [v.copy() for v in vel_hist[0]],
                'position_history': pos_hist, 'velocity_history': vel_hist}
        return video, meta

    def __getitem__(self, idx: int):
        rng = np.random.RandomState(self.seed + idx if self.seed else idx) if self.deterministic else self.rng
        video, meta = self.generate_video(rng)
        video = torch.from_numpy(video).float()
        if self.transform is not None:
            video = self.transform(video)
        return video, meta

    def generate_with_control(self, digit_labels: List[int], motion_type: str = 'random',
                              velocity_magnitude: float = 3.0):
        rng = np.random.RandomState()
        sign = lambda: 1 if rng.random() > 0.5 else -1
        vels = []
        for _ in digit_labels:
            if motion_type == 'horizontal':
                v = (velocity_magnitude * sign(), 0)
            elif motion_type == 'vertical':
                v = (0, velocity_magnitude * sign())
            elif motion_type == 'diagonal':
                v = (velocity_magnitude * 0.707 * sign(), velocity_magnitude * 0.707 * sign())
            else:
                a = rng.uniform(0, 2 * np.pi)
                v = (velocity_magnitude * np.cos(a), velocity_magnitude * np.sin(a))
            vels.append(np.array(v))
        video, meta = self.generate_video(rng, digit_labels=digit_labels, initial_velocities=vels)
        meta['motion_type'] = motion_type
        return torch.from_numpy(video).float(), meta


class MovingMNISTWithLabels(MovingMNIST):
    """Adds discretised motion labels (8 directions x 3 speeds) to the metadata."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.num_directions = 8
        self.num_speeds = 3

    def _get_motion_label(self, velocity: np.ndarray) -> dict:
        vx, vy = velocity
        mag = np.sqrt(vx ** 2 + vy ** 2)
        ang = np.arctan2(vy, vx)
        direction = int((ang + np.pi) / (2 * np.pi) * self.num_directions) % self.num_directions
        lo, hi = self.velocity_range
        third = (hi - lo) / 3
        speed = 0 if mag < lo + third else (1 if mag < lo + 2 * third else 2)
        return {'direction': direction, 'speed': speed, 'angle': ang, 'magnitude': mag}

    def __getitem__(self, idx: int):
        video, meta = super().__getitem__(idx)
        meta['motion_labels'] = [self._get_motion_label(v) for v in meta['initial_velocities']]
        meta['content_labels'] = meta['labels']
        return video, meta


def _collate_fn(batch):
    return torch.stack([b[0] for b in batch]), [b[1] for b in batch]


class DeviceClipLoader:
    """Drop-in for the DataLoader over a deterministic MovingMNIST dataset that renders the frames on
    the GPU: the host only simulates the seeded trajectories (numpy's generator, so digit choice and
    positions are those of the host path), uploads a few hundred bytes per clip and one kernel writes
    the [B,T,1,S,S] batch in device memory.  Batches are bit-identical to the host-rendered ones.

    Index order comes from the same torch samplers a DataLoader would build (RandomSampler /
    SequentialSampler / a DistributedSampler passed in), so shuffling consumes the global torch RNG
    exactly as the host path does.  Yields (clips on `device`, list of metadata dicts)."""

    def __init__(self, dataset: MovingMNIST, batch_size: int, shuffle: bool = False, sampler=None,
                 device: Optional[torch.device] = None, drop_last: bool = False, with_meta: bool = True):
        """with_meta=False skips building the per-clip metadata dictionaries (a training loop does
        not read them) and yields empty dicts in their place."""
        self.with_meta = with_meta
        if not dataset.deterministic:
            raise ValueError("DeviceClipLoader needs a deterministic dataset (per-index seeding)")
        if dataset.transform is not None:
            raise ValueError("DeviceClipLoader does not apply host-side transforms")
        from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler
        self.dataset = dataset
        self.batch_size = batch_size
        self.sampler = sampler if sampler is not None else (RandomSampler(dataset) if shuffle
                                                            else SequentialSampler(dataset))
        self.batch_sampler = BatchSampler(self.sampler, batch_size, drop_last)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        if self.device.type != 'cuda':
            raise RuntimeError("DeviceClipLoader renders on the GPU; pass a cuda device")
        self._bank = None
        self._stream = None

    def __len__(self) -> int:
        return len(self.batch_sampler)

    def render(self, indices):
        """clips [len(indices),T,1,S,S] on the device and their metadata.  The uploads and the render kernel run
        on a stream of this loader, so that the host never waits for the training stream (a copy from pageable
        memory blocks the host until everything queued before it on ITS stream is done: on the training stream
        that would be the whole step in flight); the caller's current stream is made to wait for the clips."""
        from hipvae import ops
        ds = self.dataset
        if self._bank is None:
            self._bank = torch.from_numpy(ds.digit_images).to(self.device).contiguous()
            self._stream = torch.cuda.Stream(device=self.device)
            torch.cuda.current_stream(self.device).synchronize()
        didx, ipos, metas = ds.simulate_batch(list(indices), self.with_meta)
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._stream):
            didx = torch.from_numpy(didx).to(self.device, non_blocking=True)
            pos = torch.from_numpy(ipos).to(self.device, non_blocking=True)
            clips = ops.render_clips(self._bank, didx, pos, ds.frame_size)
        cur.wait_stream(self._stream)
        clips.record_stream(cur)            # allocated on the loader's stream, consumed on the caller's
        return clips, metas

    def __iter__(self):
        # a DataLoader iterator draws its worker base seed from the global generator before its sampler
        # draws the shuffle seed; the same draw here keeps the index order (and the generator state
        # afterwards) identical to the host path
        torch.empty((), dtype=torch.int64).random_()
        for batch in self.batch_sampler:
            yield self.render(batch)


def get_moving_mnist_dataloaders(root: str = './data', batch_size: int = 32, train_size: int = 10000,
                                 val_size: int = 1000, test_size: int = 1000, sequence_length: int = 20,
                                 frame_size: int = 64, num_digits: int = 2, num_workers: int = 4,
                                 seed: Optional[int] = 42, deterministic: bool = True, with_labels: bool = False,
                                 digit_bank: Optional[np.ndarray] = None, pin_memory: bool = False,
                                 rank: int = 0, world_size: int = 1, render_device: Optional[torch.device] = None,
                                 render_meta: bool = False):
    """train / val / test loaders.  train and val draw digits from the MNIST training split (seed+0 and
    seed+1000), test from the test split (seed+2000), as in the reference (:323-354).  With
    world_size > 1 the training set is sharded round-robin over ranks (same shuffle on every rank).
    render_device: render the frames on that GPU (DeviceClipLoader) instead of in DataLoader workers;
    render_meta: also build the per-clip metadata dictionaries there (slow, and a training loop does not
    read them; without it the loader yields empty dicts)."""
    cls = MovingMNISTWithLabels if with_labels else MovingMNIST
    off = lambda o: None if seed is None else seed + o
    common = dict(root=root, sequence_length=sequence_length, frame_size=frame_size, num_digits=num_digits,
                  deterministic=deterministic, digit_bank=digit_bank)
    train_ds = cls(train=True, num_videos=train_size, seed=off(0), **common)
    val_ds = cls(train=True, num_videos=val_size, seed=off(1000), **common)
    test_ds = cls(train=False, num_videos=test_size, seed=off(2000), **common)
    sampler = None
    if world_size > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(train_ds, num_replicas=world_size, rank=rank, shuffle=True, drop_last=True)
    if render_device is not None and deterministic and not with_labels:
        # frames rendered on the GPU (not in the reference): same clips, no host rendering or H2D copy
        mk = lambda ds, shuffle, smp=None: DeviceClipLoader(ds, batch_size, shuffle=shuffle and smp is None,
                                                            sampler=smp, device=render_device,
                                                            with_meta=render_meta)
    else:
        mk = lambda ds, shuffle, smp=None: DataLoader(ds, batch_size=batch_size, shuffle=shuffle and smp is None,
                                                      sampler=smp, num_workers=num_workers, collate_fn=_collate_fn,
                                                      pin_memory=pin_memory)
    return mk(train_ds, True, sampler), mk(val_ds, False), mk(test_ds, False)
