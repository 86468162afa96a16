"""Input side of the hot path (first half): the reference's on-disk sample format and the hand-over to the
GPU.  A projected WoodScape sample is one `.npz` with `hp_img` uint8 [3, Npix] (RGB on the first `base_pix` HEALPix base
pixels, nested order) and `hp_mask` uint8 [Npix] (class ids), written at heal_swin/data/segmentation/project_on_s2.py:365-372
and read back at heal_swin/data/segmentation/hp_datasets.py:92-98.  The fisheye -> sphere projection that produces such samples is
heal_swin_amd/projection.py (`HPProjector`, HIP sampling kernels); here they are consumed (or synthesised for tests /
benchmarks) in this format.

The model takes the uint8 batch as it is: `SwinHPTransformerSys.forward` converts to the activation dtype on the GPU, so the
3-bytes-per-pixel tensor is what crosses PCIe (the reference's caller does `.float()` on the host side of the model call,
model_lightning_swin_hp.py:61 -- 4x the bytes)."""
import os

import numpy as np
import torch

IMG_KEY, MASK_KEY = "hp_img", "hp_mask"


def write_sample(path, hp_img, hp_mask):
    """One sample in the reference's format (np.savez with the two keys)."""
    hp_img, hp_mask = np.asarray(hp_img), np.asarray(hp_mask)
    assert hp_img.dtype == np.uint8 and hp_img.ndim == 2, "hp_img is uint8 [channels, Npix]"
    assert hp_mask.ndim == 1 and hp_mask.shape[0] == hp_img.shape[1], "hp_mask is [Npix]"
    np.savez(path, **{IMG_KEY: hp_img, MASK_KEY: hp_mask.astype(np.uint8)})


class HPSegmentationNpzDataset(torch.utils.data.Dataset):
    """Directory of `.npz` samples; `ds[i]` returns `(hp_img, hp_mask)` numpy arrays exactly as the reference dataset does."""

    def __init__(self, root):
        self.root = root
        self.file_names = sorted(f for f in os.listdir(root) if f.endswith(".npz"))
        self.names = [os.path.splitext(f)[0] for f in self.file_names]
        self.paths = [os.path.join(root, f) for f in self.file_names]

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, idx):
        data = np.load(self.paths[idx])
        return data[IMG_KEY], data[MASK_KEY]

    def get_item_by_name(self, name):
        return self[self.names.index(name)]


def collate_uint8(samples):
    """List of (hp_img, hp_mask) -> (uint8 [B, 3, Npix], uint8 [B, Npix]) in pinned host memory when a GPU is present."""
    imgs = torch.from_numpy(np.stack([s[0] for s in samples]))
    masks = torch.from_numpy(np.stack([s[1] for s in samples]))
    if torch.cuda.is_available():
        imgs, masks = imgs.pin_memory(), masks.pin_memory()
    return imgs, masks


class DeviceBatch(tuple):
    """(imgs, masks) on the device, as returned by `to_device`.  When the copies ran on a side stream, `ready` is the event
    recorded behind them and `wait()` must be called on the consuming stream before the tensors are used."""

    ready = None

    def wait(self, stream=None):
        """Make `stream` (default: the current stream) wait for the copies, and tell the caching allocator that the
        tensors are used there, so their blocks are not handed out again while that stream still reads them."""
        if self.ready is not None:
            stream = torch.cuda.current_stream(self[0].device) if stream is None else stream
            stream.wait_event(self.ready)
            for t in self:
                t.record_stream(stream)
        return self


def to_device(batch, device, stream=None):
    """Asynchronous host -> HBM copy of a collated uint8 batch.  Without `stream` the copies are ordered on the current
    stream and the result can be used at once.  With a side `stream` (to overlap the previous step's compute) the result
    carries the event recorded behind the copies: call `.wait()` on it before the first use --
        nxt = to_device(host_batch, dev, copy_stream); ...; imgs, masks = nxt.wait()"""
    imgs, masks = batch
    if stream is None:
        return DeviceBatch((imgs.to(device, non_blocking=True), masks.to(device, non_blocking=True)))
    with torch.cuda.stream(stream):
        out = DeviceBatch((imgs.to(device, non_blocking=True), masks.to(device, non_blocking=True)))
        out.ready = torch.cuda.Event()
        out.ready.record(stream)
    return out


def data_spec_of(sample, n_classes, class_names=None):
    """DataSpec (dim_in, f_in, f_out, base_pix) of a sample: Npix = base_pix * nside^2 with nside a power of two (when several
    factorisations exist the largest base_pix is taken: 786 432 -> 12 x 256^2, 524 288 -> 8 x 256^2)."""
    from .data_spec import DataSpec
    f_in, npix = sample[0].shape
    for bp in range(12, 0, -1):
        if npix % bp == 0:
            ns2 = npix // bp
            ns = int(round(ns2 ** 0.5))
            if ns * ns == ns2 and ns & (ns - 1) == 0:
                return DataSpec(dim_in=npix, f_in=f_in, f_out=n_classes, base_pix=bp, class_names=class_names or [])
    raise ValueError(f"{npix} pixels is not base_pix * nside^2 for any base_pix <= 12 and power-of-two nside")
