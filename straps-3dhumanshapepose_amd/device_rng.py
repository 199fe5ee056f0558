"""Counter-based random draws on the device (csrc/augment.hip: Philox4x32-10 behind `straps_philox_fill`).

The reference draws its augmentation noise from two generators -- torch's device generator (shape, camera, joint and
vertex noise) and numpy's host generator (body-part removal, occlusion boxes, crop jitter), train loop :121-175.  Here one
generator lives on the GPU: a draw is a pure function of (seed, step, sub-stream, index), the step counter is a device
int64 that `advance()` bumps with a kernel, so a captured hipGraph replays with fresh numbers and the CPU oracle can
regenerate every draw of any step (oracle/straps_oracle.py::philox_uniform / philox_normal)."""
import torch

from . import hipabi

UNIFORM, NORMAL = 0, 1


class DeviceDraws:
    def __init__(self, seed, device):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('DeviceDraws needs a GPU device (the STRAPS hot path has no CPU fallback)')
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.device)

    def fill(self, out, kind, substream):
        """fill the contiguous fp32 tensor `out` with uniform [0,1) (kind 0) or standard-normal (kind 1) draws."""
        hipabi.require_gpu_tensor(out, 'draw buffer', torch.float32)
        if not out.is_contiguous():
            raise RuntimeError('DeviceDraws.fill: the draw buffer must be contiguous')
        hipabi.check(hipabi.lib().straps_philox_fill(self.seed, hipabi.ptr(self.counter), 0, int(substream), hipabi.ptr(out), out.numel(),
                                                     int(kind), hipabi.stream_ptr()), 'straps_philox_fill')
        return out

    def uniform(self, *shape, substream=0):
        return self.fill(torch.empty(*shape, device=self.device, dtype=torch.float32), UNIFORM, substream)

    def normal(self, *shape, substream=1):
        return self.fill(torch.empty(*shape, device=self.device, dtype=torch.float32), NORMAL, substream)

    def advance(self, delta=1):
        hipabi.check(hipabi.lib().straps_counter_add(hipabi.ptr(self.counter), 1, int(delta), hipabi.stream_ptr()), 'straps_counter_add')

    def step(self):
        """host copy of the step counter (synchronises; for tests / checkpoints)."""
        return int(self.counter.item())

    def set_step(self, step):
        self.counter.fill_(int(step))


_default = {}          # (device type, index) -> (DeviceDraws, base seed it was built from | None when pinned by manual_seed(seed, device))
_seed = None          # set by manual_seed(); None: derived from torch's seed and the rank on first use


def _base_seed():
    """seed of the module-level generators: what manual_seed() set, else torch.initial_seed() + the process's rank -- like the reference's
    unseeded torch / numpy generators, every process and every run draws its own stream unless the caller seeds (torch.manual_seed or
    device_rng.manual_seed)."""
    if _seed is not None:
        return _seed
    rank = 0
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank = dist.get_rank()
    except Exception:      # noqa: BLE001
        rank = 0
    return (int(torch.initial_seed()) + rank) & 0xFFFFFFFFFFFFFFFF


def default_draws(device):
    """module-level generator of the function-style augmentation entry points (one per device, created on first use from
    `_base_seed()`; distinct devices get distinct streams: device index x an odd 64-bit constant is added modulo 2^64, so two (seed, device)
    pairs can in principle meet -- harmless, the streams are still valid).  While device_rng.manual_seed has not been called the generator
    FOLLOWS torch's seed: a torch.manual_seed() issued after the first draw rebuilds it (step counter back to 0), like the reference's
    torch-generator draws."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (device.type, idx)
    base = _base_seed()
    ent = _default.get(key)
    if ent is None or (ent[1] is not None and ent[1] != base):
        ent = (DeviceDraws(base + 0x9E3779B97F4A7C15 * idx, torch.device(device.type, idx)), base)
        _default[key] = ent
    return ent[0]


def manual_seed(seed, device=None):
    """reseed the module-level generator(s) used by augmentation.augment_* / image_utils when no draws are passed.  device = None: every
    device -- existing generators are dropped and generators created later use the same seed; a device: that device's generator only."""
    global _seed
    if device is None:
        _seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        _default.clear()
        return default_draws(torch.device('cuda', torch.cuda.current_device()))
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    _default[(device.type, idx)] = (DeviceDraws(int(seed) + 0x9E3779B97F4A7C15 * idx, torch.device(device.type, idx)), None)   # (None: pinned, ignores torch's seed)
    return _default[(device.type, idx)][0]
