"""Input side of the decode path — SURVEY.md §8 row a20: the reference's control loaders
(controlnet/utils.py:10-52) with the same names, argument meaning and error behaviour.

    read_flo                  utils.py:10-19   Middlebury .flo -> [H,W,2] float32 (pixel units)
    resize_flow_to            utils.py:21-28   bilinear (align_corners=True) + vector rescale -> [1,2,h,w]
    load_pair_to_sixch        utils.py:30-39   two RGB images (PIL, BICUBIC resize) -> [1,6,H,W] in [0,1]
    load_controls_and_flows   utils.py:41-52   -> (controlnet_cond [1,6,H,W], flow_cond [1,4,H,W]) on `device`

File I/O and the PIL resize stay on the host, as in the reference.  With a GPU `device`, `load_controls_and_flows` uploads
the raw bytes (uint8 images, the .flo payload) and does the float conversion, 6-channel packing, flow resize and vector
rescale as C-ABI launches; with `device="cpu"` it is the reference's host arithmetic.  The only
dependency dropped is torchvision (`TF.to_tensor` is restated: HWC uint8 -> CHW float32 / 255)."""
import numpy as np
import torch
import torch.nn.functional as F

FLO_MAGIC = 202021.25


def read_flo(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        magic = np.fromfile(f, np.float32, 1)[0]
        if magic != FLO_MAGIC:
            raise ValueError(f"Invalid .flo file: {path} (magic={magic})")
        w = int(np.fromfile(f, np.int32, 1)[0])
        h = int(np.fromfile(f, np.int32, 1)[0])
        data = np.fromfile(f, np.float32, 2 * w * h).reshape(h, w, 2)
    return data


def write_flo(path: str, flow_hw2: np.ndarray) -> None:
    """Inverse of read_flo (used by tests and by tools that synthesise controls)."""
    h, w, _ = flow_hw2.shape
    with open(path, "wb") as f:
        np.array([FLO_MAGIC], np.float32).tofile(f)
        np.array([w, h], np.int32).tofile(f)
        flow_hw2.astype(np.float32).tofile(f)


def resize_flow_to(flow_hw2: np.ndarray, target_h: int, target_w: int) -> torch.Tensor:
    ft = torch.from_numpy(np.ascontiguousarray(flow_hw2)).permute(2, 0, 1).unsqueeze(0)
    _, _, h, w = ft.shape
    ft = F.interpolate(ft, size=(target_h, target_w), mode="bilinear", align_corners=True)
    ft[:, 0] *= (target_w / max(w, 1))
    ft[:, 1] *= (target_h / max(h, 1))
    return ft


def _to_tensor(img) -> torch.Tensor:
    a = np.array(img, dtype=np.uint8)          # a writable copy (PIL exposes a read-only buffer)
    return torch.from_numpy(a).permute(2, 0, 1).float().div(255.0)


def _load_rgb_u8(p, size):
    from PIL import Image
    img = Image.open(p).convert("RGB")
    if size is not None:
        img = img.resize(size, Image.BICUBIC)
    return np.array(img, dtype=np.uint8)


def load_pair_to_sixch(path0, path1, size=(512, 512)) -> torch.Tensor:
    return torch.cat([_to_tensor(_load_rgb_u8(path0, size)), _to_tensor(_load_rgb_u8(path1, size))], dim=0).unsqueeze(0)


def load_controls_and_flows(img0_path, img1_path, fwd_flo_path, bwd_flo_path, size=(512, 512), device="cuda",
                            dtype=torch.float32):
    h, w = size
    if torch.device(device).type == "cuda":
        from . import ops
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device, non_blocking=True)
        sixch = ops.pack_sixch(up(_load_rgb_u8(img0_path, size)), up(_load_rgb_u8(img1_path, size)))
        flow4 = torch.empty((1, 4, h, w), device=device, dtype=torch.float32)
        ops.flow_hw2_resize_scale(up(read_flo(fwd_flo_path)), h, w, out=flow4[0, :2])
        ops.flow_hw2_resize_scale(up(read_flo(bwd_flo_path)), h, w, out=flow4[0, 2:])
        return sixch.to(dtype), flow4.to(dtype)
    sixch = load_pair_to_sixch(img0_path, img1_path, size=size).to(device=device, dtype=dtype)
    fwd_t = resize_flow_to(read_flo(fwd_flo_path), h, w)
    bwd_t = resize_flow_to(read_flo(bwd_flo_path), h, w)
    flow4 = torch.cat([fwd_t, bwd_t], dim=1).to(device=device, dtype=dtype)
    return sixch, flow4
