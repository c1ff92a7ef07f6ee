"""Tensor-level wrappers over the C ABI: borrow device pointers from torch tensors, enqueue on
torch's current HIP stream, allocate outputs with torch's caching allocator.  No arithmetic
happens in Python / ATen here -- every op below is one or more kernels of libfacodec_hip.so.
"""
import ctypes as C
import math
import os

import torch

from . import _lib
from .wprep import prepared
from ._lib import ConvDesc, VqDesc, PAD_REFLECT, PAD_ZERO, ACT_NONE, ACT_TANH, ACT_MISH, ACT_LOG_MEL, ACT_GATE, ACT_WN_RES_SKIP  # noqa: F401


def pad32(n):
    return (n + 31) & ~31


def cin_pad(c):
    """Packed weights carry zero rows up to a multiple of 48 input channels (fac_cin_pad)."""
    return ((c + 47) // 48) * 48


# torch.cuda.current_stream() builds a Stream object through five layers of Python (7 us per call, 11 000 calls = 40 ms of host time
# per train step: tools/tune/host_profile.py, round 6); the raw handle is one C call.  Same value: the current stream of the
# current device, side-stream contexts included.
_SLOW_STREAM = os.environ.get("FAC_SLOW_STREAM") == "1"           # A/B switch: torch.cuda.current_stream() as in rounds 1-5
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _raw_stream(device_index=None):
    """The current HIP stream of the (current) device as an integer handle."""
    if _RAW_STREAM is not None and _GET_DEVICE is not None and not _SLOW_STREAM:
        return _RAW_STREAM(_GET_DEVICE() if device_index is None else device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def _stream():
    return C.c_void_p(_raw_stream())


_SYNC_H2D = os.environ.get("FAC_SYNC_H2D") == "1"        # A/B switch: the blocking copies of rounds 1-5


def h2d(t, device, dtype=None):
    """Small host tensor (masks, lengths, offsets) -> device WITHOUT blocking the host: through the pinned caching allocator and an
    asynchronous copy on the current stream.  `t.to(device)` from pageable memory makes the host wait until the device has drained
    everything queued before it on that stream -- four such copies per train step (the quantizer-dropout masks) cost the step the
    whole lead of the host over the device and ~11 ms of idle GPU (tools/gpu_idle.py, round 6).  Tensors already on the device
    only change dtype."""
    if t.is_cuda:
        return t if dtype is None or t.dtype == dtype else t.to(dtype)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if torch.device(device).type != "cuda" or _SYNC_H2D:
        return t.to(device)
    staged = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    staged.copy_(t)
    return staged.to(device, non_blocking=True)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dev(t, what="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.FacodecHipError(f"{what} must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != torch.float32:
        raise _lib.FacodecHipError(f"{what} must be float32 (got {t.dtype})")
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------------- weights (K6)
@prepared
def wn_scale(v, g, out=None):
    """scale[i] = g[i]/||v[i]|| (dac/model/encodec.py:42-51)."""
    v = _dev(v, "weight_v")
    g = _dev(g, "weight_g")
    n = v.shape[0]
    scale = out if out is not None else torch.empty(n, device=v.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_wn_scale(_ptr(v), _ptr(g), _ptr(scale), n, v.numel() // n, _stream()), "fac_wn_scale")
    return scale


@prepared
def pack_conv_weight(v, g=None, out=None, scale=None):
    """(C_out, C_in, K) [+ weight-norm gain g (C_out,1,1)] -> packed (cin_pad(C_in), K, pad32(C_out));
    the rows of the padding channels are written as zeros by the kernel (no separate fill)."""
    v = _dev(v, "weight")
    if v.dim() == 2:
        v = v.unsqueeze(-1)
    c_out, c_in, k = v.shape
    cp = pad32(c_out)
    if scale is None and g is not None:
        scale = wn_scale(v, g)
    if out is None:
        out = torch.empty(cin_pad(c_in), k, cp, device=v.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_pack_conv_w(_ptr(v), _ptr(scale), _ptr(out), c_out, c_in, k, cp, _stream()),
               "fac_pack_conv_w")
    return out


@prepared
def pack_convtr_weight(v, g, stride, out=None):
    """ConvTranspose1d (C_in, C_out, 2*stride) -> polyphase packed (stride, cin_pad(C_in), 2, pad32(C_out))."""
    v = _dev(v, "weight")
    c_in, c_out, k = v.shape
    if k != 2 * stride:
        raise _lib.FacodecHipError(f"polyphase ConvTranspose1d needs kernel_size == 2*stride (got {k}, {stride})")
    cp = pad32(c_out)
    scale = wn_scale(v, g) if g is not None else None
    if out is None:
        out = torch.empty(stride, cin_pad(c_in), 2, cp, device=v.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_pack_convtr_w(_ptr(v), _ptr(scale), _ptr(out), c_in, c_out, stride, cp, _stream()),
               "fac_pack_convtr_w")
    return out


# All-phases-per-workgroup launch of the causal ConvTranspose1d (fac_conv_desc.row_phases): contiguous stores through the
# LDS epilogue instead of `stride` interleaved strided streams (2.9x the algorithmic HBM traffic).  Needs the 256-column
# tile, so it takes over from CONVTR_ROWS_MIN_T input columns; FAC_CONVTR_ROWS=0 keeps the polyphase launch everywhere.
CONVTR_ROWS = os.environ.get("FAC_CONVTR_ROWS", "0") != "0"
CONVTR_ROWS_MIN_T = 512


def convtr_rows_pad(c_out, stride):
    cpt = 128 // stride
    return -(-c_out // cpt) * 128


def convtr_rows_ok(t_in, stride, causal=True):
    return CONVTR_ROWS and causal and 2 <= stride <= 16 and t_in >= CONVTR_ROWS_MIN_T


@prepared
def pack_convtr_weight_rows(v, g, stride, out=None):
    """ConvTranspose1d (C_in, C_out, 2*stride) -> (cin_pad(C_in), 2, rows) with rows = (channel, phase) pairs in 128-row
    tiles (fac_pack_convtr_w_rows); conv_transpose1d recognises the layout by its 3 dimensions."""
    v = _dev(v, "weight")
    c_in, c_out, k = v.shape
    if k != 2 * stride:
        raise _lib.FacodecHipError(f"ConvTranspose1d needs kernel_size == 2*stride (got {k}, {stride})")
    scale = wn_scale(v, g) if g is not None else None
    if out is None:
        out = torch.empty(cin_pad(c_in), 2, convtr_rows_pad(c_out, stride), device=v.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_pack_convtr_w_rows(_ptr(v), _ptr(scale), _ptr(out), c_in, c_out, stride, _stream()),
               "fac_pack_convtr_w_rows")
    return out


def convtr_split_ok(c_in, c_out, stride, batch, t_in, causal=True, alpha_in=None):
    """All-phases ConvTranspose1d on the split-bf16 GEMM kernel (conv1d_gemm_split.hip, K = 2): mirrors conv_gsplit_ok."""
    return (BF16_SPLIT and GEMM_SPLIT and causal and alpha_in is None and 2 <= stride <= 16 and c_in >= 64
            and t_in >= 256 and batch * t_in >= 1024 and c_out * stride >= 64)


def convtr_weight_for(v, g, stride, t_in, causal=True, batch=1, alpha_in=None):
    """The packed weights conv_transpose1d wants for an input of `batch` clips of t_in columns: (split GEMM buffer, rows) for the
    bf16-pipe launch, the fp32 rows layout (opt-in) or the polyphase layout."""
    if causal and alpha_in is None and pw_taps_ok(v.shape[0], v.shape[1], 2 * stride, stride, True, batch, t_in):
        return pack_convtr_weight_rows(v, g, stride)          # stride 2, few channels: the streaming kernel with taps
    if convtr_split_ok(v.shape[0], v.shape[1], stride, batch, t_in, causal, alpha_in):
        return pack_convtr_weight_rows_split(v, g, stride)
    return pack_convtr_weight_rows(v, g, stride) if convtr_rows_ok(t_in, stride, causal) else pack_convtr_weight(v, g, stride)


# --------------------------------------------------------------------------------- conv (K1-K4)
class ConvLaunchProfile:
    """Opt-in per-launch timing of the conv kernel with HIP events recorded on the launch stream
    (bench.py's `roofline` leg).  Collects (variant name, algorithmic FLOPs, start, end)."""

    def __init__(self):
        self.records = []

    def summary(self):
        """{variant: dict(launches, flops, ms)}; call after a device synchronise."""
        out = {}
        for name, flops, e0, e1 in self.records:
            r = out.setdefault(name, dict(launches=0, flops=0.0, ms=0.0))
            r["launches"] += 1
            r["flops"] += flops
            r["ms"] += e0.elapsed_time(e1)
        return out


_PROFILE = None


class FlopCounter:
    """Algorithmic FLOPs of the GEMM-shaped work, counted where the C ABI is called (2 FLOP per multiply-add of the
    mathematical definition -- padding, tile remainders, the polyphase zero taps and the six-term bf16 split are NOT
    counted): conv forward / data gradients 2 B C_out T_out C_in K (a stride-s data gradient as the transposed conv it
    is: 2 B C_in T_in C_out K / s per ... i.e. one MAC per (output sample, tap that exists)), weight gradients
    2 B C_out C_in K T_out, LSTM recurrences 2 * 4H * H per (clip, step).  Elementwise work, FFT-free losses' small GEMMs
    booked as "dft" and left out of `total`, attention score products excluded (< 0.1 %).  Install with set_flop_counter();
    by key: conv, wgrad, lstm, dft."""

    def __init__(self):
        self.flops = dict(conv=0.0, wgrad=0.0, lstm=0.0, dft=0.0)
        self.launches = dict(conv=0, wgrad=0, lstm=0, dft=0)

    def add(self, key, f):
        self.flops[key] += f * _FLOP_SCALE
        self.launches[key] += 1

    @property
    def total(self):
        """Model MACs x 2: everything but the DFT-as-GEMM front-ends."""
        return sum(v for k, v in self.flops.items() if k != "dft")


_FLOPS = None
_FLOP_SCALE = 1.0


def set_flop_counter(c):
    global _FLOPS
    _FLOPS = c


_FLOP_KEY = "conv"


class flop_key:
    """Conv launches inside are booked under `key` instead of "conv": the windowed-DFT and mel-filterbank GEMMs of the STFT
    front-ends ("dft") are this build's way of doing an FFT, not MACs of the model."""

    def __init__(self, key):
        self.key = key

    def __enter__(self):
        global _FLOP_KEY
        self.prev, _FLOP_KEY = _FLOP_KEY, self.key

    def __exit__(self, *a):
        global _FLOP_KEY
        _FLOP_KEY = self.prev


class flop_scale:
    """Launches inside count `s` times their nominal FLOPs: the LSTM buffers carry the batch padded to 32 columns per time
    step, and the padding is not algorithmic work (s = B / BP)."""

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        global _FLOP_SCALE
        self.prev, _FLOP_SCALE = _FLOP_SCALE, self.s

    def __exit__(self, *a):
        global _FLOP_SCALE
        _FLOP_SCALE = self.prev

# k = 7 convs on the bf16 matrix pipe with fp32-grade operand splitting (conv1d_bsplit.hip).  Results have
# fp32-MFMA-grade error; FAC_BF16_SPLIT=0 keeps every conv on the fp32 MFMA kernel.
BF16_SPLIT = os.environ.get("FAC_BF16_SPLIT", "1") != "0"


def set_conv_profile(p):
    global _PROFILE
    _PROFILE = p


_CONV_WS = {}
CONV_WS_BYTES = 32 << 20


_STREAM_SLOTS = {}


def register_stream_slot(stream):
    """Gives `stream` (a torch.cuda.Stream that will carry launches concurrently with other streams: the streaming session's
    second chain, the discriminators' side streams) its own conv scratch buffer; returns the slot.  Unregistered streams --
    the caller's stream, a graph-capture stream -- share slot 0, so nothing is allocated inside a capture."""
    key = stream.cuda_stream
    if key not in _STREAM_SLOTS:
        _STREAM_SLOTS[key] = len(_STREAM_SLOTS) + 1
    return _STREAM_SLOTS[key]


_SIDE_STREAMS = {}


def side_streams(device, n):
    """n cached side streams of `device`, each with its own conv scratch (register_stream_slot)."""
    key = (device, n)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
        for st in _SIDE_STREAMS[key]:
            register_stream_slot(st)
    return _SIDE_STREAMS[key]


def join_side_streams(device):
    """The current stream waits for everything queued so far on the concurrent-chain side streams of `device`.  Used in front of a
    gradient collective launched from inside backward: autograd orders a node behind the nodes that feed it, not behind the
    parameter-gradient kernels other chains queued on their own streams, and the collective reads the whole bucket."""
    if device.type != "cuda":
        return
    cur = torch.cuda.current_stream(device)
    for (dev, _), streams in _SIDE_STREAMS.items():
        if dev.type == "cuda" and (dev.index is None or device.index is None or dev.index == device.index):
            for st in streams:
                if st != cur:
                    cur.wait_stream(st)


def run_chains(chains, device, n_streams, inputs=()):
    """Independent chains of launches (callables) side by side on up to n_streams side streams that fork from and join the
    caller's stream; returns their results in order.  Kernels with fewer workgroups than CUs (or a partly filled last round)
    then overlap with another chain's kernels; autograd replays every node's backward on the stream of its forward, so the
    backward passes of the chains overlap the same way.

    inputs: the tensors the chains READ that were allocated on the caller's stream (nested lists / dicts are walked).  They are
    recorded on every side stream in use (Tensor.record_stream), because of what happens in BACKWARD: a chain's node saves such a
    tensor, its backward runs on the side stream, and the moment that backward function returns -- its kernels merely queued --
    the engine drops the saved tensor; the caching allocator then hands the block to the next allocation on the CALLER's stream,
    which is not ordered behind the side stream at that point.  Found in round 6 as a flaky 1 - 2 % error in ONE gradient
    (timbre_encoder.spectral.0.weight: the last backward node of the timbre chain reads the log-mel features, allocated on the main
    stream and freed right after; tools/tune/race_probe.py, profiles/r06_race_probe.log).  The forward has no such window: the join
    below orders the caller's stream behind everything the chains queued."""
    n = min(n_streams, len(chains))
    if n <= 1 or device.type != "cuda":
        return [c() for c in chains]
    main = torch.cuda.current_stream(device)
    streams = side_streams(device, n)
    for st in streams:
        st.wait_stream(main)
        _record_stream(inputs, st)
    outs = []
    for j, c in enumerate(chains):
        with torch.cuda.stream(streams[j % n]):
            outs.append(c())
    for st in streams:
        main.wait_stream(st)
    # The results live in the side streams' allocator pools but are consumed (and saved for backward) by nodes of the caller's
    # stream: tell the caching allocator, or a block freed on the host while the main-stream reader is still queued could be
    # handed to another chain's side-stream kernel (the forward is covered by the fork / join above, backward frees
    # are not).
    _record_stream(outs, main)
    return outs


def _record_stream(obj, stream):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record_stream(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            _record_stream(o, stream)


_CUDA_DEVICES = {}


def _cuda_device(index):
    d = _CUDA_DEVICES.get(index)
    if d is None:
        d = _CUDA_DEVICES[index] = torch.device("cuda", index)
    return d


def _conv_workspace(device):
    """Zero-filled scratch handed to every conv launch (fac_conv_desc.ws), used by the split-reduction kernel for launches
    with few output columns: the partial sums live there between the two kernels of a launch, so it belongs to ONE stream
    at a time -- one buffer per (device, registered stream)."""
    slot = _STREAM_SLOTS.get(_raw_stream(device.index), 0) if _STREAM_SLOTS else 0
    key = (device, slot)
    ws = _CONV_WS.get(key)
    if ws is None:
        ws = _CONV_WS[key] = torch.zeros(CONV_WS_BYTES // 4, device=device, dtype=torch.float32)
    return ws


def _launch_conv(d, what):
    lib = _lib.load()
    ws = _conv_workspace(_cuda_device(_GET_DEVICE() if _GET_DEVICE is not None else torch.cuda.current_device()))
    d.ws, d.ws_bytes = ws.data_ptr(), CONV_WS_BYTES
    if _FLOPS is not None:
        _FLOPS.add(_FLOP_KEY, 2.0 * d.B * d.n_phase * max(1, d.row_phases) * d.C_out * d.T_out * d.C_in * d.K + (2.0 * d.B * d.C_out * d.T_out * d.C_out if d.w_k1 else 0.0))
    if _PROFILE is None:
        _lib.check(lib.fac_conv1d_fwd(C.byref(d), _stream()), what)
        return
    buf = C.create_string_buffer(64)
    lib.fac_conv1d_variant(C.byref(d), buf, 64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.fac_conv1d_fwd(C.byref(d), _stream()), what)
    e1.record()
    flops = 2.0 * d.B * d.n_phase * max(1, d.row_phases) * d.C_out * d.T_out * d.C_in * d.K
    if d.w_k1:
        flops += 2.0 * d.B * d.C_out * d.T_out * d.C_out   # fused 1x1 conv
    _PROFILE.records.append((buf.value.decode(), flops, e0, e1))


# 1- / 2-tap convs as split-bf16 GEMMs (conv1d_gemm_split.hip): policy of who packs the GEMM layout.  FAC_GEMM_SPLIT=0 off.
GEMM_SPLIT = os.environ.get("FAC_GEMM_SPLIT", "1") != "0"
GEMM_SPLIT_MIN_CIN = int(os.environ.get("FAC_GEMM_SPLIT_MIN_CIN", "256"))


def gemm_split_ok(c_out, c_in, k, n_cols, t_out=None):
    """1- / 2-tap stride-1 conv worth the bf16 pipe: many input channels (below ~256 the k = 1 layers are HBM-bound and the
    streaming kernel conv1d_pw.hip is the right tool), at least half a row tile, enough columns."""
    if not (BF16_SPLIT and GEMM_SPLIT and k in (1, 2)):
        return False
    if c_in < GEMM_SPLIT_MIN_CIN or c_out < 64 or n_cols < 1024:
        return False
    return k == 1 or (t_out is not None and t_out >= 256)


def gemm_split_strided_ok(c_out, c_in, k, stride, batch, t_out):
    """Strided conv with stride < k <= 2 * stride (the encoder's k = 2 s downsampling convs, the period discriminators' k = 5
    stride-3 convs) as a 2-tap split GEMM over the `stride` phase sub-signals: mirrors conv_gsplit_ok."""
    return (BF16_SPLIT and GEMM_SPLIT and 1 < stride <= 16 and stride < k <= 2 * stride and c_in >= 32
            and c_out >= 64 and t_out >= 256 and batch * t_out >= 1024)


@prepared
def pack_gemm_weight_split(v, g=None, out=None, in_stride=1, scale=None):
    """(C_out, C_in, K) [weight-normed with g over dim 0] -> fac_pack_gemm_w_split layout (uint8 buffer); K <= 2, or a strided
    conv's taps (in_stride < K <= 2 * in_stride)."""
    v = _dev(v, "weight")
    if v.dim() == 2:
        v = v.unsqueeze(-1)
    c_out, c_in, k = v.shape
    lib = _lib.load()
    if scale is None and g is not None:
        scale = wn_scale(v, g)
    nbytes = lib.fac_gemm_w_split_bytes(c_out, c_in, k, in_stride)
    if out is None:
        out = torch.empty(nbytes, device=v.device, dtype=torch.uint8)
    _lib.check(lib.fac_pack_gemm_w_split(_ptr(v), c_in * k, k, 1, _ptr(scale), out.data_ptr(), c_out, c_in, k, in_stride, _stream()),
               "fac_pack_gemm_w_split")
    return out


def pack_gemm_weight_split_t(w, out=None):
    """Materialised conv weight w (C_out, C_in, 1) -> split GEMM weights of the TRANSPOSED 1x1 (rows = C_in, contraction over
    C_out): the data gradient of a 1x1 conv, read through strides (no transposed copy)."""
    w = _dev(w, "weight")
    if w.dim() == 2:
        w = w.unsqueeze(-1)
    c_out, c_in, k = w.shape
    assert k == 1
    lib = _lib.load()
    nbytes = lib.fac_gemm_w_split_bytes(c_in, c_out, 1, 1)
    if out is None:
        out = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    _lib.check(lib.fac_pack_gemm_w_split(_ptr(w), 1, c_in, 1, None, out.data_ptr(), c_in, c_out, 1, 1, _stream()),
               "fac_pack_gemm_w_split(transposed)")
    return out


BSPLIT2 = os.environ.get("FAC_BSPLIT2", "1") != "0"


def split2_ok(c_out, k, k1, stride, n_cols):
    """Few-output-channel 9- / 3-tap conv (plain or two-level, taps per level k1) for conv1d_bsplit2.hip: mirrors
    conv_bsplit2_ok."""
    kv = k1 if 0 < k1 < k else k
    return (BF16_SPLIT and BSPLIT2 and 8 <= c_out <= 32 and ((kv == 9 and stride in (1, 2)) or (kv == 3 and stride == 1))
            and n_cols >= 4096)


@prepared
def pack_conv_weight_split2(v, g=None, k1=0, out=None, scale=None):
    """(C_out <= 32, C_in, K) [weight-normed with g] -> fac_pack_conv_w_split2 layout (uint8 buffer); k1: taps per level."""
    v = _dev(v, "weight")
    c_out, c_in, k = v.shape
    lib = _lib.load()
    if scale is None and g is not None:
        scale = wn_scale(v, g)
    nbytes = lib.fac_conv_w_split2_bytes(c_out, c_in, k, k1)
    if out is None:
        out = torch.empty(nbytes, device=v.device, dtype=torch.uint8)
    _lib.check(lib.fac_pack_conv_w_split2(_ptr(v), _ptr(scale), out.data_ptr(), c_out, c_in, k, k1, _stream()),
               "fac_pack_conv_w_split2")
    return out


@prepared
def pack_convtr_weight_rows_split(v, g, stride, out=None):
    """ConvTranspose1d (C_in, C_out, 2*stride) -> split GEMM weights of the all-phases launch: the (channel, phase) rows of
    pack_convtr_weight_rows, each row's (C_in, 2) taps as bf16 planes.  Returns (split buffer, rows)."""
    rows = pack_convtr_weight_rows(v, g, stride)                  # (cin_pad, 2, R) fp32, weight-norm applied
    cip, _, R = rows.shape
    c_in = v.shape[0]
    lib = _lib.load()
    nbytes = lib.fac_gemm_w_split_bytes(R, c_in, 2, 1)
    if out is None:
        out = torch.empty(nbytes, device=v.device, dtype=torch.uint8)
    _lib.check(lib.fac_pack_gemm_w_split(_ptr(rows), 1, 2 * R, R, None, out.data_ptr(), R, c_in, 2, 1, _stream()),
               "fac_pack_gemm_w_split(convtr rows)")
    return out, R


@prepared
def pack_conv_weight_split(v, g=None, out=None, scale=None):
    """(C_out, C_in, K) [weight-normed with g] -> split-bf16 layout of fac_pack_conv_w_split (K = 5 / 7; uint8 buffer) or of
    fac_pack_gemm_w_split (K = 1 / 2)."""
    v = _dev(v, "weight")
    c_out, c_in, k = v.shape
    if k <= 2:
        return pack_gemm_weight_split(v, g, out=out, scale=scale)
    lib = _lib.load()
    if scale is None and g is not None:
        scale = wn_scale(v, g)
    nbytes = lib.fac_conv_w_split_bytes(c_out, c_in, k)
    if out is None:
        out = torch.empty(nbytes, device=v.device, dtype=torch.uint8)
    _lib.check(lib.fac_pack_conv_w_split(_ptr(v), _ptr(scale), out.data_ptr(), c_out, c_in, k, _stream()),
               "fac_pack_conv_w_split")
    return out


def conv_out_len(t_in, k, stride, dilation):
    """Output length and (pad_left, extra_right) of a causal SConv1d (dac/model/encodec.py:71-78,212-222)."""
    k_eff = (k - 1) * dilation + 1
    padding_total = k_eff - stride
    n_frames = (t_in - k_eff + padding_total) / stride + 1
    ideal = (math.ceil(n_frames) - 1) * stride + (k_eff - padding_total)
    extra = ideal - t_in
    t_out = (t_in + padding_total + extra - k_eff) // stride + 1
    return t_out, padding_total, extra


class P8:
    """An activation tensor (B, C, T) as three bf16 planes hi / mid / lo (hi + mid + lo == value exactly), each laid out
    [b][c / 8][t][8] -- the operand format of the split-bf16 kernels' LDS stages (fac_conv_desc.x_p8): `planes` is a (3, B, C / 8, T, 8)
    bfloat16 tensor; every plane ends with one zero unit (what consumers read for padding columns), so `planes` is stored as
    (3, B * C / 8 * T + 1, 8) bfloat16."""

    def __init__(self, planes, shape):
        self.planes, self.shape = planes, tuple(shape)

    @staticmethod
    def empty(B, C, T, device):
        return P8(torch.empty(3, B * (C // 8) * T + 1, 8, device=device, dtype=torch.bfloat16), (B, C, T))

    @property
    def plane_bytes(self):
        return self.planes.stride(0) * 2

    def to_float(self):
        """The fp32 tensor the planes add up to (tests)."""
        B, C, T = self.shape
        v = self.planes[:, :-1].float().sum(0).reshape(B, C // 8, T, 8)     # exact: three addends of disjoint significance
        return v.permute(0, 1, 3, 2).reshape(B, C, T).contiguous()


def to_p8(x, alpha=None):
    """fp32 (B, C, T) [-> snake(x, alpha)] -> P8 (fac_to_p8)."""
    x = _dev(x, "x")
    B, C, T = x.shape
    out = P8.empty(B, C, T, x.device)
    _lib.check(_lib.load().fac_to_p8(_ptr(x), _ptr(alpha), _ptr(out.planes), B, C, T, _stream()), "fac_to_p8")
    return out


# Inference launches of the split GEMM kernel whose every input byte feeds at least this many FLOPs take their input as P8
# planes made by ONE extra pass over it (fac_to_p8: 4 bytes read, 6 written per element at ~6 TB/s): the kernel then stages both
# operands by LDS-DMA instead of splitting fp32 values in its staging waves (+11 .. 20 % on such a launch), which
# pays for the pass when the input is small next to the GEMM -- the LSTM input projections, the transposed convs with many output
# rows per input sample, the last two strided convs of the encoder (threshold sweep 150 / 300 / 450 / 700 on one box: 56.3 / 56.0 /
# 55.5 / 56.0 ms per forward).  0 disables.
P8_PREPASS_MIN_FLOP_PER_BYTE = float(os.environ.get("FAC_P8_PREPASS", "450"))


def p8_prepass(x, flop_per_in_byte):
    """x (B, C, T) fp32 -> P8 when the policy above says the pass pays for itself (inference only: no autograd node), else x."""
    if (P8_PREPASS_MIN_FLOP_PER_BYTE <= 0 or flop_per_in_byte < P8_PREPASS_MIN_FLOP_PER_BYTE or not BF16_SPLIT or isinstance(x, P8)
            or torch.is_grad_enabled() or not x.is_cuda or x.shape[1] % 8 != 0 or x.shape[0] * (x.shape[1] // 8) * x.shape[2] * 16 >= (1 << 32)):
        return x
    return to_p8(x)


def conv1d(x, w_packed, c_out, k, bias=None, stride=1, dilation=1, pad_left=None, pad_mode=PAD_REFLECT,
           t_out=None, alpha_in=None, alpha_out=None, res=None, act=ACT_NONE, out=None, causal=True,
           alpha_y2=None, want_y=True, w_k1=None, bias_k1=None, w_split=None, k1=0, dilation2=0, skip_acc=None):
    """Fused conv (see fac_conv1d_fwd).  x (B, C_in, T).  With pad_left=None the SConv1d padding
    rule is applied (causal: everything on the left; non-causal: asymmetric split).
    k1 / dilation2: two-level taps (tap k = k2 * k1 + k1' reads offset k2 * dilation2 + k1' * dilation), see fac_conv_desc.
    alpha_y2: also produce y2 = snake(y, alpha_y2) (returned as (y, y2); y is None if not want_y).
    w_k1 / bias_k1: fused ResidualUnit tail -- y = w_k1 * snake(conv + bias, alpha_out) + bias_k1 + res.
    act=ACT_GATE / ACT_WN_RES_SKIP (streaming hops only, facodec_hip.h): the output has c_out / 2 channels; ACT_WN_RES_SKIP adds
    the first half of the channels to `res` (-> out, which may be res itself) and the second half onto `skip_acc` in place."""
    x_p8 = x if isinstance(x, P8) else None
    if x_p8 is not None:
        B, c_in, t_in = x_p8.shape
    else:
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.stride(2) == 1):
            x = _dev(x, "x")     # channel-sliced views (time contiguous) are consumed in place
        B, c_in, t_in = x.shape
    dev_ = x_p8.planes.device if x_p8 is not None else x.device
    if pad_left is None:
        t_o, padding_total, _ = conv_out_len(t_in, k, stride, dilation)
        pad_left = padding_total if causal else padding_total - padding_total // 2
        if t_out is None:
            t_out = t_o
    if t_out is None:
        raise ValueError("t_out required with explicit pad_left")
    cp = w_packed.shape[-1] if w_packed is not None else pad32(c_out)
    c_y = c_out // 2 if act in (ACT_GATE, ACT_WN_RES_SKIP) else c_out
    if out is None and want_y:
        out = torch.empty(B, c_y, t_out, device=dev_, dtype=torch.float32)
    y2 = torch.empty(B, c_out, t_out, device=dev_, dtype=torch.float32) if alpha_y2 is not None else None
    if act == ACT_WN_RES_SKIP:
        assert skip_acc is not None and skip_acc.is_contiguous() and skip_acc.shape == out.shape and alpha_y2 is None
        y2 = skip_acc
    res = _dev(res, "res")
    d = ConvDesc()
    d.bias = bias.data_ptr() if bias is not None else None
    if x_p8 is not None:
        d.x, d.x_p8, d.x_p8_plane_bytes = None, x_p8.planes.data_ptr(), x_p8.plane_bytes
    else:
        d.x = x.data_ptr()
    d.w = w_packed.data_ptr() if w_packed is not None else w_split.data_ptr()   # split-only launch: see fac_conv_desc.w_split
    d.alpha_in = alpha_in.data_ptr() if alpha_in is not None else None
    d.alpha_out = alpha_out.data_ptr() if alpha_out is not None else None
    d.res = res.data_ptr() if res is not None else None
    d.y = out.data_ptr() if out is not None else None
    d.y2 = y2.data_ptr() if y2 is not None else None
    d.alpha_y2 = alpha_y2.data_ptr() if alpha_y2 is not None else None
    d.w_k1 = w_k1.data_ptr() if w_k1 is not None else None
    d.bias_k1 = bias_k1.data_ptr() if bias_k1 is not None else None
    d.w_split = w_split.data_ptr() if w_split is not None else None
    d.x_bs, d.x_cs = (x.stride(0), x.stride(1)) if x_p8 is None else (c_in * t_in, t_in)
    d.y_bs, d.y_cs = c_y * t_out, t_out
    if out is not None and (out.stride(0), out.stride(1)) != (d.y_bs, d.y_cs) and out.shape[0] * out.shape[1] > 1:
        # a time-contiguous view of a wider buffer as the output (streaming: the LSTM pre-activations of the real batch columns)
        assert out.stride(2) == 1 and y2 is None and res is None and out.shape == (B, c_y, t_out)
        d.y_bs, d.y_cs = out.stride(0), out.stride(1)
    d.B, d.C_in, d.T_in, d.C_out, d.C_out_pad, d.T_out = B, c_in, t_in, c_out, cp, t_out
    d.K, d.stride, d.dilation, d.pad_left, d.pad_mode = k, stride, dilation, pad_left, pad_mode
    d.n_phase, d.y_tstride, d.act, d.w_batched, d.w_bs = 1, 1, act, 0, 0
    d.K1, d.dilation2 = k1, dilation2
    d.pw_split = 1 if (BF16_SPLIT and PW_SPLIT and (k == 1 or (PW_TAPS and k == 4 and stride == 2))) else 0
    _launch_conv(d, "fac_conv1d_fwd")
    return (out, y2) if alpha_y2 is not None else out


# Streaming hops: the WaveNet's elementwise launches folded into the reduction kernels of its convs (FAC_STREAM_FOLD=0: separate
# launches, the round-5 hop; bit-identical either way -- tests/test_gpu_parity.py::test_streaming_folded_epilogues_are_bit_identical)
STREAM_FOLD = os.environ.get("FAC_STREAM_FOLD", "1") != "0"
SKINNY_MAX_COLS = 640


def conv_transpose1d(x, w_packed, c_out, stride, bias=None, alpha_in=None, out=None, alpha_y2=None, causal=True,
                     has_history=False):
    """Causal SConvTranspose1d (kernel 2*stride, right trim k-stride: dac/model/encodec.py:248-270)
    as `stride` polyphase 2-tap convs: y[., t*s+p] = W[p] x[t] + W[p+s] x[t-1].
    has_history (streaming): x's first column is x[t0-1] of an earlier chunk (instead of the zero of
    the start of the signal); x may then be a time-contiguous view of a wider buffer."""
    x_p8 = x if isinstance(x, P8) else None
    if x_p8 is not None:       # all-phases split-GEMM launch only (fac_conv1d_fwd rejects P8 operands anywhere else)
        assert not has_history
        B, c_in, t_in = x_p8.shape
        x_bs, x_cs = c_in * t_in, t_in
    else:
        if not (has_history and x.is_cuda and x.dtype == torch.float32 and x.stride(2) == 1):
            x = _dev(x, "x")
        B, c_in, t_in = x.shape
        x_bs, x_cs = x.stride(0), x.stride(1)
    dev_ = x_p8.planes.device if x_p8 is not None else x.device
    t_in_full = t_in
    if has_history:
        assert causal
        t_in -= 1
    t_total = t_in * stride
    split_rows = isinstance(w_packed, tuple)       # (split GEMM buffer, rows) from pack_convtr_weight_rows_split
    cp = w_packed[1] if split_rows else w_packed.shape[-1]
    if out is None:
        out = torch.empty(B, c_out, t_total, device=dev_, dtype=torch.float32)
    d = ConvDesc()
    d.bias = bias.data_ptr() if bias is not None else None
    if x_p8 is not None:
        d.x, d.x_p8, d.x_p8_plane_bytes = None, x_p8.planes.data_ptr(), x_p8.plane_bytes
    else:
        d.x = x.data_ptr()
    if split_rows:
        d.w, d.w_split = None, w_packed[0].data_ptr()
    else:
        d.w = w_packed.data_ptr()
    d.alpha_in = alpha_in.data_ptr() if alpha_in is not None else None
    d.alpha_out, d.res, d.y = None, None, out.data_ptr()
    y2 = torch.empty_like(out) if alpha_y2 is not None else None
    d.y2 = y2.data_ptr() if y2 is not None else None
    d.alpha_y2 = alpha_y2.data_ptr() if alpha_y2 is not None else None
    d.x_bs, d.x_cs = x_bs, x_cs
    d.y_bs, d.y_cs = c_out * t_total, t_total
    d.B, d.C_in, d.T_in, d.C_out, d.C_out_pad, d.T_out = B, c_in, t_in_full, c_out, cp, t_in
    d.K, d.stride, d.dilation, d.pad_left, d.pad_mode = 2, 1, 1, (0 if has_history else 1), PAD_ZERO
    d.n_phase, d.y_tstride, d.act, d.w_batched, d.w_bs = stride, stride, ACT_NONE, 0, 0
    # non-causal: trim ceil(s/2) on the left, floor(s/2) on the right (dac/model/encodec.py:265-269)
    d.phase_shift = 0 if causal else stride - stride // 2
    if split_rows or w_packed.dim() == 3:          # rows layouts: all phases per workgroup, contiguous stores
        if not causal or has_history:
            raise _lib.FacodecHipError("the all-phases ConvTranspose1d launch is causal and offline only")
        d.n_phase, d.y_tstride, d.phase_shift, d.row_phases = 1, 1, 0, stride
        d.pw_split = 1 if (BF16_SPLIT and PW_SPLIT and PW_TAPS and not split_rows and stride == 2) else 0
    _launch_conv(d, "fac_conv1d_fwd(convtr)")
    return (out, y2) if alpha_y2 is not None else out


# Short clips on the split GEMM kernel as ONE flattened signal (layers.SConv1d._run_flat / SConvTranspose1d.run, round 4, inference):
# the same trick for the TRAINING launches (round 6).  Per-clip column tiles of that kernel would be half empty at the 160-frame
# latent rate, so those launches (the encoder's last downsampling conv, the decoder's first ConvTranspose1d, and their data
# gradients) fell to the fp32 128 x 160 tile at 38 - 63 TFLOP/s (profiles/r06_train_layers_serial.log: 3.9 ms per step).
FLAT_TRAIN = os.environ.get("FAC_FLAT_TRAIN", "1") != "0"
# k = 1 ResidualUnit tails at C = 64 .. 384 on the bf16 pipe inside the streaming kernel (conv1d_pw_split.hip); 0: fp32 MFMAs (rounds 2-5)
PW_SPLIT = os.environ.get("FAC_PW_SPLIT", "1") != "0"


def pw_split_tail_ok(c_in, c_out, cols):
    """C = 256 / 384 ResidualUnit tails (and their data gradients) of the TRAINING step on the streaming bf16-plane kernel instead
    of the split GEMM kernel (round 6: 0.29 -> 0.2 ms per launch at 16 x 4800 columns)."""
    return BF16_SPLIT and PW_SPLIT and c_in == c_out and c_in in (256, 384) and cols >= 65536


# stride-2 layers with few channels on the streaming kernel with taps (conv1d_pw_split.hip, conv1d_pwt_kernel); follows PW_SPLIT
PW_TAPS = os.environ.get("FAC_PW_TAPS", "1") != "0"


def pw_taps_ok(c_in, c_out, k, stride, transposed, batch, t_out):
    """Mirror of conv_pwt_ok: the causal ConvTranspose1d with stride 2 (all output phases as rows, fp32 weights of
    pack_convtr_weight_rows) or a k = 4 stride-2 conv (fp32 weights of pack_conv_weight) whose C_in * taps <= 384 virtual channels
    fit the LDS as bf16 planes for 64 output rows; t_out = output columns per clip at the INPUT rate (transposed) / output rate."""
    if not (BF16_SPLIT and PW_SPLIT and PW_TAPS and stride == 2):
        return False
    taps, rows = (2, 2 * c_out) if transposed else (4, c_out)
    if k != 4 or (c_in * taps) % 64 or c_in * taps > 384 or rows % 64:
        return False
    return batch * ((t_out + 31) // 32) >= 2 * (256 // (rows // 64)) * 12


def flat_strided_ok(c_out, c_in, k, s, batch, n_out):
    """A k = 2 s strided conv over `batch` clips of n_out outputs each, too short for per-clip tiles but long enough as one signal."""
    return (FLAT_TRAIN and k == 2 * s and s > 1 and n_out < 256 and not gemm_split_strided_ok(c_out, c_in, k, s, batch, n_out)
            and gemm_split_strided_ok(c_out, c_in, k, s, 1, batch * (n_out + 1) - 1))


def conv1d_flat_strided(xp, w_split, c_out, k, s, bias=None):
    """xp (B, C_in, (n + 1) s): every clip already carries its s columns of padding (reflected on the left for the causal forward
    conv, zeros on the right for the data gradient of a transposed conv).  The clips are laid one after another as ONE signal and
    the strided conv runs WITHOUT padding: output column b (n + 1) + t is output t of clip b (same products, same order as the
    per-clip launch), plus one junk column per clip where the window straddles two clips, dropped on the way back -> (B, C_out, n)."""
    B, c_in, L = xp.shape
    n = L // s - 1
    xf = xp.permute(1, 0, 2).reshape(1, c_in, B * L)
    t_out = B * (n + 1) - 1
    xf = p8_prepass(xf, 2.0 * c_out * k / (4.0 * s))
    y = conv1d(xf, None, c_out, k, bias=bias, stride=s, pad_left=0, pad_mode=PAD_ZERO, t_out=t_out, w_split=w_split)
    full = torch.empty(c_out, B * (n + 1), device=y.device, dtype=y.dtype)
    full[:, :t_out] = y[0]
    return full.reshape(c_out, B, n + 1)[:, :, :n].permute(1, 0, 2).contiguous()


def flat_convtr_ok(c_in, c_out, s, batch, t_cols):
    """An all-phases ConvTranspose1d launch over `batch` clips of t_cols input columns each (incl. their zero column), short clips."""
    return (FLAT_TRAIN and t_cols < 256 and not convtr_split_ok(c_in, c_out, s, batch, t_cols)
            and convtr_split_ok(c_in, c_out, s, 1, batch * t_cols))


def conv_transpose1d_flat(xz, v, g, s, bias=None):
    """xz (B, C_in, T'): clips that already hold their zero column (in FRONT for the forward transposed conv: the x[t - 1] of a
    clip's first frame; at the END for the data gradient of a strided conv, where it is the next clip's x[t - 1] as well).  One
    signal of B T' columns through the all-phases split-GEMM launch -> (B, C_out, T' s): per clip exactly the per-clip launch's
    output (y[t s + p] = W[p] x[t] + W[p + s] x[t - 1])."""
    B, c_in, T1 = xz.shape
    c_out = v.shape[1]
    xf = xz.permute(1, 0, 2).reshape(1, c_in, B * T1)
    xf = p8_prepass(xf, 2.0 * c_out * 2 * s / 4.0)
    y = conv_transpose1d(xf, pack_convtr_weight_rows_split(v, g, s), c_out, s, bias=bias, causal=True)
    return y.reshape(c_out, B, T1 * s).permute(1, 0, 2).contiguous()


def snake(x, alpha, out=None):
    x = _dev(x, "x")
    B, c, t = x.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.load().fac_snake_fwd(_ptr(x), _ptr(alpha), _ptr(out), B, c, t, _stream()), "fac_snake_fwd")
    return out


# --------------------------------------------------------------------------------- LSTM (K5)
def lstm_to_time_major(x):
    x = _dev(x, "x")
    B, H, T = x.shape
    xT = torch.empty(H, T, pad32(B), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_lstm_to_time_major(_ptr(x), _ptr(xT), B, H, T, _stream()), "fac_lstm_to_time_major")
    return xT


def lstm_from_time_major(yT, skip, B, alpha=None):
    H, T, BP = yT.shape
    out = torch.empty(B, H, T, device=yT.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_lstm_from_time_major(_ptr(yT), _ptr(skip), _ptr(alpha), _ptr(out), B, H, T, _stream()),
               "fac_lstm_from_time_major")
    return out


def pack_lstm_whh(w_hh, out=None):
    w_hh = _dev(w_hh, "weight_hh")
    H = w_hh.shape[1]
    if out is None:
        out = torch.empty_like(w_hh)
    _lib.check(_lib.load().fac_pack_lstm_whh(_ptr(w_hh), _ptr(out), H, _stream()), "fac_pack_lstm_whh")
    return out


LSTM_PERSIST = os.environ.get("FAC_LSTM_PERSIST", "1") != "0"
LSTM_PERSIST_MAX_BATCH = int(os.environ.get("FAC_LSTM_PERSIST_MAX_BATCH", "16"))


_LSTM_ARMED = set()


def _lstm_arm():
    """The resident kernels' waits give up after 4 s without progress and count that in a host-mapped word the device has to be
    told about (fac_lstm_persist_arm: one one-thread kernel, once per device, outside any stream capture).  A device whose word is
    not armed answers fac_lstm_persist_ok = 0, so the word is armed here, eagerly, the first time anybody asks."""
    dev = torch.cuda.current_device()
    if dev in _LSTM_ARMED:
        return
    if torch.cuda.is_current_stream_capturing():
        return                                   # not now; the per-step kernels take this call, a later eager call arms
    if _lib.load().fac_lstm_persist_arm(_stream()):
        _LSTM_ARMED.add(dev)


def lstm_timeouts():
    """Resident-LSTM waits that gave up in this process so far (host-mapped counter: no synchronisation; it counts what the device
    has EXECUTED, so a caller that wants the verdict on launches it just queued synchronises first)."""
    return int(_lib.load().fac_lstm_persist_timeouts())


def raise_if_lstm_timed_out(what="resident LSTM launch"):
    """Results of a resident launch whose wait gave up are meaningless.  Training is protected on the device (optim.FlatAdamW: the
    abort flag travels with the per-parameter flags and the masked AdamW step steps nothing on ANY rank); this is the host-side
    report for inference callers and benchmarks, to be called behind a synchronisation point."""
    n = lstm_timeouts()
    if n:
        raise _lib.FacodecHipError(f"{what}: {n} resident-LSTM wait(s) timed out on this device -- the outputs of that launch are "
                                   "meaningless (later layers use the per-step kernels: fac_lstm_persist_ok answers 0 from now on)")


def lstm_persist_ok(H, batch):
    """True when the one-launch-per-layer recurrence (lstm_persist.hip) is used for an (H, batch) zero-initial-state
    layer on this device: H in {512, 1024, 1536}, H/8 workgroups <= CUs, batch <= 16.  The kernels cover batch <= 32
    (two 16-column blocks) but measure no faster than the per-step launches there (profiles/r03_lstm_bench.log)."""
    if not (LSTM_PERSIST and batch is not None and batch <= LSTM_PERSIST_MAX_BATCH) or _ranks_share_a_device():
        return False
    _lstm_arm()
    return (bool(_lib.load().fac_lstm_persist_ok(int(H), int(batch)))
            and bool(_lib.load().fac_lstm_persist_stream_ok(_stream())))


LSTM_PERSIST_SPLIT = os.environ.get("FAC_LSTM_PERSIST_SPLIT", "1") != "0"
LSTM_PERSIST_SPLIT_MIN_BATCH = int(os.environ.get("FAC_LSTM_PERSIST_SPLIT_MIN_BATCH", "17"))


LSTM_PERSIST_SPLIT_MAX_SCRATCH = int(float(os.environ.get("FAC_LSTM_PERSIST_SPLIT_MAX_SCRATCH_GB", "1")) * 2 ** 30)


def lstm_persist_split_ok(H, batch, T=None):
    """True when an inference layer runs on the resident kernel with bf16 x 3 operands (lstm_persist.hip,
    fac_lstm_layer_fwd_persist_split): 17 .. 32 batch columns, where the fp32 resident kernel is bound by its fp32 MFMA time and
    the per-step kernel by re-streaming W_hh.  Follows FAC_BF16_SPLIT like the conv kernels of the same arithmetic.
    T (frames): the kernel needs one fresh exchange region per step, T * H * 32 * 6 bytes of scratch per layer call -- 47 MB at the
    benchmark's 160 frames, but 1.5 - 3 GB for 30-second clips; above FAC_LSTM_PERSIST_SPLIT_MAX_SCRATCH_GB (1) the per-step kernels,
    which need none, take the layer."""
    if T is not None and int(T) * int(H) * 32 * 6 > LSTM_PERSIST_SPLIT_MAX_SCRATCH:
        return False
    if not (LSTM_PERSIST and LSTM_PERSIST_SPLIT and BF16_SPLIT and batch is not None
            and LSTM_PERSIST_SPLIT_MIN_BATCH <= batch <= 32) or _ranks_share_a_device():
        return False
    _lstm_arm()
    return (bool(_lib.load().fac_lstm_persist_split_ok(int(H), int(batch)))
            and bool(_lib.load().fac_lstm_persist_stream_ok(_stream())))


def pack_lstm_whh_split(w_hh):
    """W_hh (4H, H) as the three-plane bf16 A fragments of the split resident kernel (fac_pack_lstm_whh_split)."""
    w_hh = _dev(w_hh, "weight_hh")
    out = torch.empty(w_hh.numel() * 6, device=w_hh.device, dtype=torch.uint8)
    _lib.check(_lib.load().fac_pack_lstm_whh_split(_ptr(w_hh), _ptr(out), w_hh.shape[1], _stream()), "fac_pack_lstm_whh_split")
    return out


def lstm_layer_persist_split(pre, w_hh, H, batch):
    """The whole inference layer in ONE launch on the bf16 matrix pipe: pre (4H, T, BP), raw W_hh (4H, H) -> yT (H, T, BP)."""
    _, T, BP = pre.shape
    if _FLOPS is not None:
        _FLOPS.add("lstm", 2.0 * 4 * H * H * T * BP)
    yT = torch.empty(H, T, BP, device=pre.device, dtype=torch.float32)
    if BP > 32:
        yT[..., 32:].zero_()
    hsplit = torch.empty(T * H * 32 * 6, device=pre.device, dtype=torch.uint8)       # one fresh exchange region per step
    _lib.check(_lib.load().fac_lstm_layer_fwd_persist_split(_ptr(pre), _ptr(pack_lstm_whh_split(w_hh)), _ptr(hsplit), _ptr(yT),
                                                            T, H, batch, BP, _stream()), "fac_lstm_layer_fwd_persist_split")
    return yT


def _ranks_share_a_device():
    """The resident LSTM grids of two PROCESSES on one GPU could each hold part of the CUs and spin on workgroups that never get
    one (launches are only serialised inside a process, lstm_persist.hip): with more local ranks than visible devices (the
    two-ranks-on-one-GPU smoke mode of tools/ddp_smoke.py) the per-step kernels are used instead."""
    try:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    except ValueError:
        local_world = 1
    return local_world > max(1, torch.cuda.device_count())


def pack_lstm_whh16(w_hh, transposed=False):
    """W_hh (4H, H) as the register fragments of the resident kernels (fac_pack_lstm_whh16)."""
    w_hh = _dev(w_hh, "weight_hh")
    out = torch.empty_like(w_hh)
    _lib.check(_lib.load().fac_pack_lstm_whh16(_ptr(w_hh), _ptr(out), w_hh.shape[1], 1 if transposed else 0, _stream()),
               "fac_pack_lstm_whh16")
    return out


def _zero_pad_cols(t, nc):
    """Columns >= nc of a (rows, T, BP) work buffer are batch padding the resident kernels never touch."""
    if nc < t.shape[-1]:
        t[..., nc:].zero_()


def lstm_layer_persist(pre, w_hh, H, batch, save=None):
    """The whole layer in ONE launch (fac_lstm_layer_fwd_persist): pre (4H, T, BP), raw W_hh (4H, H) -> yT (H, T, BP)."""
    _, T, BP = pre.shape
    nc = 16 * ((batch + 15) // 16)
    if _FLOPS is not None:
        _FLOPS.add("lstm", 2.0 * 4 * H * H * T * BP)
    yT = torch.empty(H, T, BP, device=pre.device, dtype=torch.float32)
    _zero_pad_cols(yT, nc)
    sg, sc = save if save is not None else (None, None)
    if save is not None:
        _zero_pad_cols(sg, nc)
        _zero_pad_cols(sc, nc)
    hfrag = torch.empty(T * H * nc, device=pre.device, dtype=torch.float32)       # one fresh exchange region per step
    _lib.check(_lib.load().fac_lstm_layer_fwd_persist(_ptr(pre), _ptr(pack_lstm_whh16(w_hh)), _ptr(hfrag), _ptr(yT), _ptr(sg),
                                                      _ptr(sc), T, H, batch, BP, _stream()), "fac_lstm_layer_fwd_persist")
    return yT


def lstm_layer(pre, whh_packed, H, state=None, step0=0, save=None):
    """pre (4H, T, BP) -> yT (H, T, BP).  state (3, H, BP): carried cell / hidden buffers of a streaming
    session with `step0` steps already taken (None: fresh zero state).  save = (gates (4H,T,BP), c (H,T,BP)):
    training mode, the activations back-propagation through time needs are stored there."""
    _, T, BP = pre.shape
    if _FLOPS is not None:
        _FLOPS.add("lstm", 2.0 * 4 * H * H * T * BP)
    yT = torch.empty(H, T, BP, device=pre.device, dtype=torch.float32)
    if state is None:
        state = torch.empty(3, H, BP, device=pre.device, dtype=torch.float32)   # cell state + 2 fragment-ordered h
        step0 = 0
    sg, sc = save if save is not None else (None, None)
    _lib.check(_lib.load().fac_lstm_layer_fwd_train(_ptr(pre), _ptr(whh_packed), _ptr(yT), _ptr(state), _ptr(sg), _ptr(sc),
                                                    T, H, BP, step0, _stream()), "fac_lstm_layer_fwd")
    return yT


def stream_push(buf, src, hist, n_prev):
    """buf (B, C, cap) left-context buffer; appends src (B, C, n) behind the last `hist` columns."""
    src = _dev(src, "src")
    B, c, cap = buf.shape
    assert buf.is_contiguous() and src.shape[:2] == buf.shape[:2]
    _lib.check(_lib.load().fac_stream_push(_ptr(buf), _ptr(src), B * c, cap, hist, n_prev, src.shape[-1], _stream()),
               "fac_stream_push")


# --------------------------------------------------------------------------------- VQ (K7)
def vq_step(z_in, w_in_packed, b_in, codebook, w_out, w_out_scale, b_out, codes_out, residual=None, zq_acc=None,
            zq_out=None, mask=None, z_e=None, loss_part=None):
    """One VectorQuantize.forward (dac/nn/quantize.py:34-70) + RVQ bookkeeping (:173-193).
    codes_out: int64 view (B, T) (may be a strided slice of a (B, N, T) tensor)."""
    z_in = _dev(z_in, "z")
    B, D, T = z_in.shape
    assert codes_out.dtype == torch.int64 and codes_out.stride(-1) == 1
    d = VqDesc()
    d.residual = residual.data_ptr() if residual is not None else None
    d.z_in = z_in.data_ptr()
    d.zq_acc = zq_acc.data_ptr() if zq_acc is not None else None
    d.zq_out = zq_out.data_ptr() if zq_out is not None else None
    d.w_in, d.b_in, d.codebook = w_in_packed.data_ptr(), b_in.data_ptr(), codebook.data_ptr()
    d.w_out, d.b_out = w_out.data_ptr(), b_out.data_ptr()
    d.w_out_scale = w_out_scale.data_ptr() if w_out_scale is not None else None
    d.mask = mask.data_ptr() if mask is not None else None
    d.codes = codes_out.data_ptr()
    d.z_e = z_e.data_ptr() if z_e is not None else None
    d.loss_part = loss_part.data_ptr() if loss_part is not None else None
    d.codes_bs = codes_out.stride(0)
    d.B, d.D, d.T, d.Kc = B, D, T, codebook.shape[0]
    _lib.check(_lib.load().fac_vq_fwd(C.byref(d), _stream()), "fac_vq_fwd")


def vq_loss_tiles(T):
    """Row length of `vq_step`'s loss_part buffer (one partial sum per time tile of the kernel)."""
    return int(_lib.load().fac_vq_loss_tiles(int(T)))


def vq_search(latents, codebook):
    """latents (N, 8) -> int64 (N,) nearest normalised code (dac/nn/quantize.py:78-94)."""
    latents = _dev(latents, "latents")
    codebook = _dev(codebook, "codebook")
    n = latents.shape[0]
    idx = torch.empty(n, device=latents.device, dtype=torch.int64)
    _lib.check(_lib.load().fac_vq_search(_ptr(latents), _ptr(codebook), _ptr(idx), n, codebook.shape[0], _stream()),
               "fac_vq_search")
    return idx


# --------------------------------------------------------------------------------- small fused ops
def gate_tanh_sigmoid(a, g=None):
    """g: optional (B, 2C) conditioning rows (may be a strided slice of a wider (B, n) tensor)."""
    a = _dev(a)
    B, c2, T = a.shape
    out = torch.empty(B, c2 // 2, T, device=a.device, dtype=torch.float32)
    g_bs = g.stride(0) if g is not None else 0
    _lib.check(_lib.load().fac_gate_tanh_sigmoid(_ptr(a), _ptr(g), g_bs, _ptr(out), B, c2 // 2, T, _stream()),
               "fac_gate_tanh_sigmoid")
    return out


def embed_sum(codes, tables, code_row0=0, out=None):
    """out (B, E, T) (+)= sum_i tables[i][codes[:, code_row0 + i]]; tables (n, V, E), codes (B, N, T) int64."""
    B, n_codes, T = codes.shape
    n_tab, V, E = tables.shape
    acc = out is not None
    if out is None:
        out = torch.empty(B, E, T, device=codes.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_embed_sum(_ptr(codes.contiguous()), _ptr(_dev(tables)), _ptr(out), B, n_tab, n_codes,
                                         code_row0, V, E, T, 1 if acc else 0, _stream()), "fac_embed_sum")
    return out


def glu_residual(a, res):
    a, res = _dev(a), _dev(res)
    B, c2, T = a.shape
    out = torch.empty(B, c2 // 2, T, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_glu_residual(_ptr(a), _ptr(res), _ptr(out), B, c2 // 2, T, _stream()), "fac_glu_residual")
    return out


def add(a, b, out=None):
    a, b = _dev(a), _dev(b)
    if out is None:
        out = torch.empty_like(a)
    _lib.check(_lib.load().fac_add(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), "fac_add")
    return out


def sub2(a, b, c, out=None):
    a, b, c = _dev(a), _dev(b), _dev(c)
    if out is None:
        out = torch.empty_like(a)
    _lib.check(_lib.load().fac_sub2(_ptr(a), _ptr(b), _ptr(c), _ptr(out), a.numel(), _stream()), "fac_sub2")
    return out


def mul_mask_(x, mask):
    """x (B,C,T) *= mask (B,T) in place."""
    B, c, T = x.shape
    _lib.check(_lib.load().fac_mul_mask(_ptr(x), _ptr(mask), B, c, T, _stream()), "fac_mul_mask")
    return x


def wn_res_skip_(rs, x, out, last):
    B, c, T = out.shape
    _lib.check(_lib.load().fac_wn_res_skip(_ptr(rs), _ptr(x), _ptr(out), B, c, T, 1 if last else 0, _stream()),
               "fac_wn_res_skip")


def attention(q, k, v, mask, n_heads):
    q, k, v = _dev(q), _dev(k), _dev(v)
    B, c, T = q.shape
    out = torch.empty_like(q)
    _lib.check(_lib.load().fac_attention(_ptr(q), _ptr(k), _ptr(v), _ptr(mask), _ptr(out), B, n_heads, c // n_heads, T,
                                         _stream()), "fac_attention")
    return out


def masked_mean(x, mask):
    x = _dev(x)
    B, c, T = x.shape
    out = torch.empty(B, c, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_masked_mean(_ptr(x), _ptr(mask), _ptr(out), B, c, T, _stream()), "fac_masked_mean")
    return out


def layernorm_c_affine(x, style):
    x, style = _dev(x), _dev(style)
    B, c, T = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.load().fac_layernorm_c_affine(_ptr(x), _ptr(style), _ptr(out), B, c, T, _stream()),
               "fac_layernorm_c_affine")
    return out


# --------------------------------------------------------------------------------- STFT pieces
def stft_frames(wave, n_win, n_frames, hop, pad, n_off):
    wave = _dev(wave, "wave")
    B, T = wave.shape
    frames = torch.empty(B, n_win, n_frames, device=wave.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_stft_frames(_ptr(wave), _ptr(frames), B, T, n_win, n_frames, hop, pad, n_off, _stream()),
               "fac_stft_frames")
    return frames


def spec_power(spec, power):
    B, f2, nf = spec.shape
    out = torch.empty(B, f2 // 2, nf, device=spec.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_spec_power(_ptr(spec), _ptr(out), B, f2 // 2, nf, power, _stream()), "fac_spec_power")
    return out


def reduce_pair(a, b, out, scratch, mode, eps=0.0, scale=1.0, accumulate=False):
    a, b = _dev(a), _dev(b)
    _lib.check(_lib.load().fac_reduce_pair(_ptr(a), _ptr(b), _ptr(out), _ptr(scratch), a.numel(), mode, eps, scale,
                                           1 if accumulate else 0, _stream()), "fac_reduce_pair")
    return out


def logdiff_rms(a, b, out, scratch, eps, scale=1.0, accumulate=False):
    """losses.py:84 term on (B, M, T) tensors, accumulated into out[0]."""
    a, b = _dev(a), _dev(b)
    B, M, T = a.shape
    _lib.check(_lib.load().fac_logdiff_rms(_ptr(a), _ptr(b), _ptr(out), _ptr(scratch), B, M, T, eps, scale,
                                           1 if accumulate else 0, _stream()), "fac_logdiff_rms")
    return out


def logdiff_rms_bwd(a, b, db, eps, scale=1.0, accumulate=False):
    """db (+)= scale * d/db sum_{b,t} sqrt(mean_m (log(|a|+eps) - log(|b|+eps))^2); a, b, db (B, M, T)."""
    a, b = _dev(a), _dev(b)
    B, M, T = a.shape
    _lib.check(_lib.load().fac_logdiff_rms_bwd(_ptr(a), _ptr(b), _ptr(db), B, M, T, eps, scale, 1 if accumulate else 0, _stream()),
               "fac_logdiff_rms_bwd")


def aa_snakebeta(x, alpha_log, beta_log, filter12):
    """Anti-aliased SnakeBeta (Activation1d(SnakeBeta(alpha_logscale=True)))."""
    x = _dev(x)
    B, c, T = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.load().fac_aa_snakebeta_fwd(_ptr(x), _ptr(alpha_log), _ptr(beta_log), _ptr(filter12), _ptr(y), B, c, T,
                                                _stream()), "fac_aa_snakebeta_fwd")
    return y


# --------------------------------------------------------------------------------- backward of the conv stack
@prepared
def flipped_weight(v, g=None, scale=None, out=None):
    """(C_out, C_in, K) [weight-normed] -> (C_in, C_out, K) weights of the data-gradient conv (channels swapped, taps flipped),
    one launch (fac_flip_transpose_w)."""
    v = _dev(v, "weight")
    c_out, c_in, k = v.shape
    if scale is None and g is not None:
        scale = wn_scale(v, g)
    if out is None:
        out = torch.empty(c_in, c_out, k, device=v.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_flip_transpose_w(_ptr(v), _ptr(scale), _ptr(out), c_out, c_in, k, _stream()), "fac_flip_transpose_w")
    return out


@prepared
def pack_conv_weight_bwd(v, g=None, scale=None, out=None):
    """(C_out, C_in, K) [weight-normed] -> packed weights of the bwd-data conv (taps flipped, channels swapped)."""
    v = _dev(v, "weight")
    c_out, c_in, k = v.shape
    packed = out if out is not None else torch.empty(cin_pad(c_out), k, pad32(c_in), device=v.device, dtype=torch.float32)
    if scale is None and g is not None:
        scale = wn_scale(v, g)
    _lib.check(_lib.load().fac_pack_conv_w_bwd(_ptr(v), _ptr(scale), _ptr(packed), c_out, c_in, k, pad32(c_in), _stream()),
               "fac_pack_conv_w_bwd")
    return packed


def conv1d_bwd_data(dy, v, g, t_in, stride=1, dilation=1, pad_mode=PAD_REFLECT, causal=True, scale=None, allow_view=False):
    """Gradient w.r.t. the input of SConv1d (dac/model/encodec.py:212-228) given dy (B, C_out, T_out).
    scale: the weight-norm scale g / ||v|| if the caller already has it (the forward computed it).
    allow_view: the caller's consumer takes rows with a stride (snake_bwd_fused): the result may then be the window
    [pad_left, pad_left + t_in) of the padded gradient rows -- a non-contiguous (B, C, t_in) view, mirrored edge samples added in
    place (fac_pad_fold_edges) -- instead of a copy (fac_pad_fold_bwd: 8 bytes per element to move the tensor by <= 54 samples)."""
    dy = _dev(dy, "dy")
    if scale is None and g is not None:
        scale = wn_scale(v, g)
    c_out, c_in, k = v.shape
    B, _, t_out = dy.shape
    t_o, padding_total, extra = conv_out_len(t_in, k, stride, dilation)
    assert t_o == t_out, (t_o, t_out)
    pad_left = padding_total if causal else padding_total - padding_total // 2
    pad_right = (padding_total - pad_left) + extra
    tp = pad_left + t_in + pad_right
    if (stride == 1 and BF16_SPLIT and c_in % 16 == 0 and c_out % 16 == 0 and B * tp > 640
            and (k == 7 or (k in (3, 5) and c_out >= 64 and c_in > 32))):
        # the flipped / transposed conv on the bf16 pipe too: materialise w = g v/||v||, swap channels, flip taps
        wt = flipped_weight(v, g, scale)                                   # (C_in, C_out, K) = weights of the bwd conv
        dxpad = conv1d(dy, None, c_in, k, dilation=dilation, pad_left=(k - 1) * dilation, pad_mode=PAD_ZERO, t_out=tp,
                       w_split=pack_conv_weight_split(wt))
    elif stride == 1 and k == 1 and gemm_split_ok(c_in, c_out, 1, B * tp) and not pw_split_tail_ok(c_in, c_out, B * tp):
        w = rows_fma(v, scale) if g is not None else v                    # 1x1: the transposed GEMM on the bf16 pipe
        dxpad = conv1d(dy, None, c_in, 1, pad_left=0, pad_mode=PAD_ZERO, t_out=tp, w_split=pack_gemm_weight_split_t(w))
    elif stride == 1:
        dxpad = conv1d(dy, pack_conv_weight_bwd(v, g, scale), c_in, k, dilation=dilation, pad_left=(k - 1) * dilation,
                       pad_mode=PAD_ZERO, t_out=tp)
    else:
        if k != 2 * stride or dilation != 1:
            raise NotImplementedError("strided bwd_data is built for the model's k = 2*stride convs")
        dy_ext = torch.cat([dy, torch.zeros(B, c_out, 1, device=dy.device)], dim=2)
        if flat_convtr_ok(c_out, c_in, stride, B, t_out + 1):      # short clips: one flattened signal (the trailing zero column of a
            dxpad = conv_transpose1d_flat(dy_ext, v, g, stride)    # clip is the x[t - 1] of the next clip's first frame)
        else:
            dxpad = conv_transpose1d(dy_ext, convtr_weight_for(v, g, stride, dy_ext.shape[-1], batch=B), c_in, stride)
        assert dxpad.shape[-1] == tp, (dxpad.shape, tp)
    if tp == t_in and FOLD_IN_PLACE in (1, 2):
        return dxpad                                  # no padding (the 1x1 convs): the padded gradient IS the gradient
    if allow_view and FOLD_IN_PLACE in (1, 3) and dxpad.is_contiguous() and (pad_mode == PAD_ZERO or t_in > pad_left + pad_right):
        if pad_mode == PAD_REFLECT:
            _lib.check(_lib.load().fac_pad_fold_edges(_ptr(dxpad), B, c_in, t_in, tp, pad_left, _stream()), "fac_pad_fold_edges")
        return dxpad[:, :, pad_left:pad_left + t_in]
    dx = torch.empty(B, c_in, t_in, device=dy.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_pad_fold_bwd(_ptr(dxpad), _ptr(dx), B, c_in, t_in, tp, pad_left, pad_mode, _stream()),
               "fac_pad_fold_bwd")
    return dx


# 1 (default): both shortcuts; 0: neither (rounds 1-5); 2: only "no padding -> no copy"; 3: only the in-place edge fold + row-stride view
FOLD_IN_PLACE = int(os.environ.get("FAC_FOLD_IN_PLACE", "1"))


def _rows_view(t):
    """(tensor, row stride) for a (B, C, T) fp32 CUDA tensor whose rows are time-contiguous and evenly spaced (a window of a wider
    buffer, see conv1d_bwd_data(allow_view=True)); anything else is made contiguous first."""
    if t is None:
        return None, 0
    if (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.stride(2) == 1 and t.stride(1) >= t.shape[2]
            and t.stride(0) == t.shape[1] * t.stride(1)):
        return t, t.stride(1)
    t = _dev(t, "dy")
    return t, t.shape[2]


def conv1d_bwd_weight(x, dy, k, stride=1, dilation=1, pad_mode=PAD_REFLECT, causal=True, pad_left=None, k1=0, dilation2=0,
                      want_db=False):
    """dW (C_out, C_in, K) of SConv1d (or of a plain conv when pad_left is given explicitly; k1 / dilation2: two-level taps).
    want_db: also return the bias gradient sum_{b,t} dy -- folded into the split kernel's pass over dy where that kernel runs."""
    x, dy = _dev(x, "x"), _dev(dy, "dy")
    B, c_in, t_in = x.shape
    _, c_out, t_out = dy.shape
    if pad_left is None:
        _, padding_total, _ = conv_out_len(t_in, k, stride, dilation)
        pad_left = padding_total if causal else padding_total - padding_total // 2
    dw = torch.empty(c_out, c_in, k, device=x.device, dtype=torch.float32)
    db = torch.empty(c_out, device=x.device, dtype=torch.float32) if want_db else None
    fused = _bwd_weight_launch(x, dy, dw, B, c_in, t_in, c_out, t_out, k, stride, dilation, pad_left, pad_mode, k1, dilation2, db=db)
    if want_db:
        return dw, (db if fused else bias_grad(dy))
    return dw


# One grow-only scratch per device for the split weight-gradient kernel (bf16 planes of both operands + partial sums): every
# launch goes to torch's current stream in program order, so consecutive layers can share it -- no per-layer allocation of
# hundreds of MB, and its size shows up once in peak_mem instead of as allocator churn.  FAC_WGRAD_WS_GB caps it.
_WGRAD_WS = {}
WGRAD_K1_STREAM = os.environ.get("FAC_WGRAD_K1_STREAM", "1") != "0"      # k = 1 tails with few channels on conv1d_wgrad_k1.hip
WGRAD_WS_CAP = int(float(os.environ.get("FAC_WGRAD_WS_GB", "6")) * (1 << 30))


def _wgrad_workspace(device, nbytes):
    """One grow-only buffer per (device, stream): the discriminators' backward passes run on several streams at once
    (discriminator.py), and a launch's operand planes / partial sums must not be overwritten by another stream's launch."""
    key = (device, _raw_stream(device.index))
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            del _WGRAD_WS[key]
            ws = None
        ws = _WGRAD_WS[key] = torch.empty((nbytes + (64 << 20)) // 4, device=device, dtype=torch.float32)
    return ws


def _bwd_weight_launch(x, dy, dw, B, c_in, t_in, c_out, t_out, k, stride, dilation, pad_left, pad_mode, k1=0, dilation2=0, db=None):
    """Weight gradient of one conv (see _bwd_weight_launch_inner); counted by the FLOP counter and, under a ConvLaunchProfile,
    timed as ONE record per conv: operand planes + GEMM + split-K reduction together."""
    flops = 2.0 * B * c_out * c_in * k * t_out
    if _FLOPS is not None:
        _FLOPS.add("wgrad", flops)
    if _PROFILE is None:
        return _bwd_weight_launch_inner(x, dy, dw, B, c_in, t_in, c_out, t_out, k, stride, dilation, pad_left, pad_mode, k1, dilation2, db)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = _bwd_weight_launch_inner(x, dy, dw, B, c_in, t_in, c_out, t_out, k, stride, dilation, pad_left, pad_mode, k1, dilation2, db)
    e1.record()
    split = BF16_SPLIT and _lib.load().fac_conv1d_bwd_weight_split_ws_bytes(B, c_in, t_in, c_out, t_out, k, stride, dilation, k1, dilation2) > 0
    _PROFILE.records.append(("weight gradient: split_planes + conv1d_wgrad_split/kmajor + reduce (bf16x3 split)" if split
                             else "weight gradient: conv1d_wgrad_kernel (fp32 MFMA) + reduce", flops, e0, e1))
    return r


def _bwd_weight_launch_inner(x, dy, dw, B, c_in, t_in, c_out, t_out, k, stride, dilation, pad_left, pad_mode, k1=0, dilation2=0, db=None):
    """dW on the bf16 matrix pipe with fp32-grade splitting (conv1d_wgrad_split.hip) when the shape qualifies and
    FAC_BF16_SPLIT is on, else on the fp32 MFMA kernel.  db: optional (C_out) buffer for the bias gradient; returns True when the
    launch filled it."""
    lib = _lib.load()
    if WGRAD_K1_STREAM and BF16_SPLIT and k == 1 and stride == 1 and pad_left == 0 and t_in == t_out and k1 in (0, 1):
        # few channels, long signal (the ResidualUnit tails at C = 64 / 96 / 192): both tensors once through the fp32 matrix pipe
        # instead of two operand-split passes + plane reads (conv1d_wgrad_k1.hip); the bias gradient rides on the dy fragments
        nb = lib.fac_conv1d_bwd_weight_k1_ws_bytes(B, c_in, c_out, t_in)
        if nb > 0:
            ws = _wgrad_workspace(x.device, nb)
            _lib.check(lib.fac_conv1d_bwd_weight_k1(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), nb, B, c_in, c_out, t_in, _stream()),
                       "fac_conv1d_bwd_weight_k1")
            return db is not None
    if WGRAD_K1_STREAM and BF16_SPLIT and k > 1 and stride == 1 and c_in * k <= 64 and c_out in (32, 64) and t_out >= 4096:
        # first layers (1 -> 64 k = 7 of the encoder, 2 -> 32 (3, 9) of the multi-resolution discriminator): the C_in * K columns of dW
        # are shifted views of one or two input rows -- the same kernel with virtual rows (fac_conv1d_bwd_weight_taps) on the padded input
        kk1 = k1 if 0 < k1 < k else k
        d2 = dilation2 if 0 < k1 < k else 0
        nb = lib.fac_conv1d_bwd_weight_taps_ws_bytes(B, c_in, c_out, t_out, k, kk1, dilation, d2)
        if nb > 0:
            tx = lib.fac_conv1d_bwd_weight_taps_tx(t_out, k, kk1, dilation, d2)
            need = (t_out - 1) + (k // kk1 - 1) * d2 + (kk1 - 1) * dilation + 1          # positions of the padded input the outputs read
            right = max(0, need - pad_left - t_in)
            if pad_mode == PAD_REFLECT and max(pad_left, right) > 0:
                xp = torch.nn.functional.pad(x, (pad_left, right), mode="reflect")         # data movement only (a 1- / 2-channel signal)
            else:
                xp = torch.nn.functional.pad(x, (pad_left, right))
            if xp.shape[-1] < tx:
                xp = torch.nn.functional.pad(xp, (0, tx - xp.shape[-1]))
            ws = _wgrad_workspace(x.device, nb)
            _lib.check(lib.fac_conv1d_bwd_weight_taps(_ptr(xp), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), nb, B, c_in, xp.shape[-1], c_out,
                                                      t_out, k, kk1, dilation, d2, _stream()), "fac_conv1d_bwd_weight_taps")
            return db is not None
    nbytes = lib.fac_conv1d_bwd_weight_split_ws_bytes(B, c_in, t_in, c_out, t_out, k, stride, dilation, k1, dilation2) if BF16_SPLIT else -1
    if nbytes > WGRAD_WS_CAP:       # beyond the workspace budget: the fp32 kernel (no operand planes) takes the layer
        nbytes = -1
    if nbytes > 0:
        ws = _wgrad_workspace(x.device, nbytes)
        if db is not None and lib.fac_conv1d_bwd_weight_split_db_ok(B, c_in, t_in, c_out, t_out, k, stride, dilation, k1, dilation2):
            _lib.check(lib.fac_conv1d_bwd_weight_split_db(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), nbytes, B, c_in, t_in, c_out,
                                                          t_out, k, stride, dilation, pad_left, pad_mode, k1, dilation2, _stream()),
                       "fac_conv1d_bwd_weight_split_db")
            return True
        _lib.check(lib.fac_conv1d_bwd_weight_split(_ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), nbytes, B, c_in, t_in, c_out, t_out, k,
                                                   stride, dilation, pad_left, pad_mode, k1, dilation2, _stream()),
                   "fac_conv1d_bwd_weight_split")
        return False
    kk1 = k1 if 0 < k1 < k else k                      # the fp32 kernel sees k / kk1 virtual channels per input channel
    nbytes = lib.fac_conv1d_bwd_weight_ws_bytes(B, c_in * (k // kk1), c_out, t_out, kk1)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    _lib.check(lib.fac_conv1d_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), nbytes, B, c_in, t_in, c_out, t_out, k,
                                         stride, dilation, pad_left, pad_mode, k1, dilation2, _stream()), "fac_conv1d_bwd_weight")
    return False


def weight_norm_bwd(v, g, dw):
    v, g, dw = _dev(v), _dev(g), _dev(dw)
    n = v.shape[0]
    dv, dg = torch.empty_like(v), torch.empty_like(g)
    _lib.check(_lib.load().fac_weight_norm_bwd(_ptr(v), _ptr(g), _ptr(dw), _ptr(dv), _ptr(dg), n, v.numel() // n, _stream()),
               "fac_weight_norm_bwd")
    return dv, dg


def snake_bwd(x, alpha, dy):
    x, dy = _dev(x, "x"), _dev(dy, "dy")
    B, c, t = x.shape
    dx, dalpha = torch.empty_like(x), torch.empty(c, device=x.device, dtype=torch.float32)
    scratch = torch.empty(32 * c, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_snake_bwd(_ptr(x), _ptr(alpha), _ptr(dy), _ptr(dx), _ptr(dalpha), _ptr(scratch), B, c, t,
                                         _stream()), "fac_snake_bwd")
    return dx, dalpha


def snake_bwd_fused(x, alpha, dy, add=None, want_bias=False):
    """Snake backward with the fan-in add and the producing conv's bias gradient fused in (fac_snake_bwd_fused):
    dx = add + dy * dsnake/dx, dalpha, [dbias = sum over (b, t) of dx]."""
    x, add = _dev(x, "x"), _dev(add, "add")
    dy, dy_rs = _rows_view(dy)              # dy may be a window of padded gradient rows (conv1d_bwd_data(allow_view=True))
    B, c, t = x.shape
    dx, dalpha = torch.empty_like(x), torch.empty(c, device=x.device, dtype=torch.float32)
    db = torch.empty(c, device=x.device, dtype=torch.float32) if want_bias else None
    scratch = torch.empty(64 * c, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_snake_bwd_fused_rs(_ptr(x), _ptr(alpha), _ptr(dy), dy_rs, _ptr(add), _ptr(dx), _ptr(dalpha), _ptr(db),
                                                  _ptr(scratch), B, c, t, _stream()), "fac_snake_bwd_fused")
    return dx, dalpha, db


def bias_grad(dy):
    dy = _dev(dy, "dy")
    B, c, t = dy.shape
    db = torch.empty(c, device=dy.device, dtype=torch.float32)
    scratch = torch.empty(32 * c, device=dy.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_bias_grad(_ptr(dy), _ptr(db), _ptr(scratch), B, c, t, _stream()), "fac_bias_grad")
    return db


def conv_transpose1d_bwd(x, dy, v, g, stride):
    """Causal SConvTranspose1d (kernel 2*stride): -> (dx, dW (C_in, C_out, K)).  dx is the strided forward conv of dy
    on the same weights; dW the weight-gradient kernel with the roles of input and output swapped."""
    x, dy = _dev(x, "x"), _dev(dy, "dy")
    B, c_in, t_in = x.shape
    c_out, k = v.shape[1], v.shape[2]
    assert k == 2 * stride and dy.shape == (B, c_out, t_in * stride)
    if pw_taps_ok(c_out, c_in, k, stride, False, B, t_in):         # stride 2, few channels: the streaming kernel with taps (fp32 pack)
        dx = conv1d(dy, pack_conv_weight(v, g), c_in, k, stride=stride, pad_left=0, pad_mode=PAD_ZERO, t_out=t_in)
    elif gemm_split_strided_ok(c_in, c_out, k, stride, B, t_in):   # the strided conv of dy on the split GEMM kernel
        dx = conv1d(dy, None, c_in, k, stride=stride, pad_left=0, pad_mode=PAD_ZERO, t_out=t_in,
                    w_split=pack_gemm_weight_split(v, g, in_stride=stride))
    elif flat_strided_ok(c_in, c_out, k, stride, B, t_in):         # short clips: zeros on the right of every clip, one flattened signal
        dx = conv1d_flat_strided(torch.nn.functional.pad(dy, (0, stride)), pack_gemm_weight_split(v, g, in_stride=stride), c_in, k, stride)
    else:
        dx = conv1d(dy, pack_conv_weight(v, g), c_in, k, stride=stride, pad_left=0, pad_mode=PAD_ZERO, t_out=t_in)
    dw = torch.empty(c_in, c_out, k, device=x.device, dtype=torch.float32)
    _bwd_weight_launch(dy, x, dw, B, c_out, t_in * stride, c_in, t_in, k, stride, 1, 0, PAD_ZERO)
    return dx, dw


def lstm_layer_bwd(d_out, w_hh, gates, cs, H, batch=None):
    """BPTT of one layer (fac_lstm_layer_bwd): d_out (H, T, BP) gradient of the layer's h sequence -> dgates (4H, T, BP).
    batch: number of real columns; with it the resident kernel (fac_lstm_layer_bwd_persist) runs where it applies."""
    _, T, BP = d_out.shape
    lib = _lib.load()
    if lstm_persist_ok(H, batch):
        dgates = torch.empty(4 * H, T, BP, device=d_out.device, dtype=torch.float32)
        nc = 16 * ((batch + 15) // 16)
        _zero_pad_cols(dgates, nc)
        scratch = torch.empty((4 + 4 * T) * H * nc, device=d_out.device, dtype=torch.float32)
        _lib.check(lib.fac_lstm_layer_bwd_persist(_ptr(_dev(d_out)), _ptr(pack_lstm_whh16(w_hh, transposed=True)), _ptr(gates),
                                                  _ptr(cs), _ptr(dgates), _ptr(scratch), T, H, batch, BP, _stream()),
                   "fac_lstm_layer_bwd_persist")
        return dgates
    wt = torch.empty(4 * H * H, device=d_out.device, dtype=torch.float32)
    _lib.check(lib.fac_pack_lstm_whh_t(_ptr(_dev(w_hh)), _ptr(wt), H, _stream()), "fac_pack_lstm_whh_t")
    dgates = torch.empty(4 * H, T, BP, device=d_out.device, dtype=torch.float32)
    scratch = torch.empty(13 * H * BP, device=d_out.device, dtype=torch.float32)
    _lib.check(lib.fac_lstm_layer_bwd(_ptr(_dev(d_out)), _ptr(wt), _ptr(gates), _ptr(cs), _ptr(dgates), _ptr(scratch), T, H, BP,
                                      _stream()), "fac_lstm_layer_bwd")
    return dgates


def lstm_gate_bwd(dy_t, rec, gates_t, c_t, c_prev, dc, dgates_t, H, BP, rs, first):
    """Views into (rows, T, BP) buffers at one time step (row stride rs); rec / dc dense (H, BP)."""
    _lib.check(_lib.load().fac_lstm_gate_bwd(_ptr(dy_t), _ptr(rec), _ptr(gates_t), _ptr(c_t), _ptr(c_prev), _ptr(dc),
                                             _ptr(dgates_t), H, BP, rs, 1 if first else 0, _stream()), "fac_lstm_gate_bwd")


def tanh_bwd(y, dy):
    y, dy = _dev(y), _dev(dy)
    dx = torch.empty_like(dy)
    _lib.check(_lib.load().fac_tanh_bwd(_ptr(y), _ptr(dy), _ptr(dx), dy.numel(), _stream()), "fac_tanh_bwd")
    return dx


def pair_bwd(a, b, da, mode, eps, scale, accumulate):
    a, b = _dev(a), _dev(b)
    _lib.check(_lib.load().fac_pair_bwd(_ptr(a), _ptr(b), _ptr(da), a.numel(), mode, eps, scale, 1 if accumulate else 0,
                                        _stream()), "fac_pair_bwd")
    return da


def spec_power_bwd(spec, dout, power):
    spec, dout = _dev(spec), _dev(dout)
    B, f2, nf = spec.shape
    dspec = torch.empty_like(spec)
    _lib.check(_lib.load().fac_spec_power_bwd(_ptr(spec), _ptr(dout), _ptr(dspec), B, f2 // 2, nf, power, _stream()),
               "fac_spec_power_bwd")
    return dspec


def stft_frames_bwd(dframes, T, hop, pad, n_off):
    dframes = _dev(dframes)
    B, n_win, nf = dframes.shape
    dwave = torch.empty(B, T, device=dframes.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_stft_frames_bwd(_ptr(dframes), _ptr(dwave), B, T, n_win, nf, hop, pad, n_off, _stream()),
               "fac_stft_frames_bwd")
    return dwave


# --------------------------------------------------------------------------------- quantizer / LayerNorm backward
def vq_latent_bwd(z_e, codebook, codes, d_zst=None, wc=None, want_dze=True, want_zst=False):
    z_e = _dev(z_e)
    B, _, T = z_e.shape
    d_ze = torch.empty_like(z_e) if want_dze else None
    z_st = torch.empty_like(z_e) if want_zst else None
    _lib.check(_lib.load().fac_vq_latent_bwd(_ptr(z_e), _ptr(codebook), C.c_void_p(codes.data_ptr()), codes.stride(0),
                                             _ptr(d_zst), _ptr(wc), _ptr(d_ze), _ptr(z_st), B, T, _stream()), "fac_vq_latent_bwd")
    return d_ze, z_st


def vq_codebook_grad(z_e, codebook, codes, wb):
    z_e = _dev(z_e)
    B, _, T = z_e.shape
    dcb = torch.empty_like(codebook)
    _lib.check(_lib.load().fac_vq_codebook_grad(_ptr(z_e), _ptr(codebook), C.c_void_p(codes.data_ptr()), codes.stride(0),
                                                _ptr(wb), _ptr(dcb), B, T, codebook.shape[0], 0, _stream()), "fac_vq_codebook_grad")
    return dcb


def layernorm_c_affine_bwd(x, style, dout):
    x, style, dout = _dev(x), _dev(style), _dev(dout)
    B, c, T = x.shape
    dx, dstyle = torch.empty_like(x), torch.empty_like(style)
    stats = torch.empty(B * T * 2, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().fac_layernorm_c_affine_bwd(_ptr(x), _ptr(style), _ptr(dout), _ptr(dx), _ptr(dstyle), _ptr(stats),
                                                      B, c, T, _stream()), "fac_layernorm_c_affine_bwd")
    return dx, dstyle


def rows_fma(a, w=None, c=None, sign=1.0):
    """a[b] * w[b] + sign * c[b] for (B, ...) tensors (w (B,), c like a; both optional)."""
    a = _dev(a)
    B = a.shape[0]
    out = torch.empty_like(a)
    _lib.check(_lib.load().fac_rows_fma(_ptr(a), _ptr(w), _ptr(_dev(c)), _ptr(out), B, a.numel() // B, sign, _stream()),
               "fac_rows_fma")
    return out
