"""Streaming causal inference (BASELINE.json configs[4]): encode -> FA-quantize -> decode in
480-sample hops with carried state, matching the OFFLINE causal model on the same signal.

The reference has no streaming code (its README only says the causal model "supports" it,
/root/reference README.md:105-107); the contract here is therefore: for every frame / sample the session
emits, codes equal and waveform within fp32 noise of `model.encoder -> model.quantizer -> model.decoder` run
on the whole signal (tests/test_gpu_parity.py::test_streaming_matches_offline).

How the offline arithmetic is reproduced incrementally
  * every causal conv (dac/model/encodec.py:212-228) owns a left-context buffer `[ (k-1)*d history | new ]`
    (`_Tap`): a hop appends its new input columns (fac_stream_push) and the SAME conv kernel runs over the
    window that the new outputs need -- no padding, so per-output arithmetic (accumulation order included) is
    the offline kernel's.  Strided convs keep their unconsumed remainder inside the same history.
  * the FIRST chunk ("prime") is run with the offline reflect padding on the left: the reference's causal convs
    reflect-pad the start of the signal, so sample 0 of the output depends on up to 2 750 later samples
    (k7, dilation 9 at 1/50 rate).  Priming therefore takes >= 4 800 samples (a multiple of 2 400 so that the
    8-frames-per-5-hops pattern starts aligned).
  * LSTMs carry (h, c) (fac_lstm_layer_fwd_from).
  * prosody branch: the centred 2 048-point STFT of frame f needs samples up to 300 f + 1 023, so the quantizer
    and decoder run 1 024 samples (2.1 hops) behind the encoder; latents wait in a small FIFO.  The timbre vector
    is fixed for the session (enrolment clip), as in any streaming use of the model.
  * a period of 5 hops (2 400 samples = 8 frames) repeats every launch geometry exactly, so hops 5..9 are
    captured into five HIP graphs and replayed from then on (launch-bound: ~450 small launches per hop).
  * `finish()` flushes the frames that were waiting for look-ahead, with the reflect padding of the END of the
    signal the offline front-end applies.
"""
import os

import torch

from . import ops
from .dac_model import FUSED_RU_CHANNELS, DecoderBlock, EncoderBlock
from .layers import ConvWeights

LSTM_REAL_COLUMNS = os.environ.get("FAC_STREAM_LSTM_REAL_COLUMNS", "1") != "0"     # _LSTMState.run
HOP = 480            # samples per streaming hop (20 ms @ 24 kHz)
FRAME = 300          # encoder hop (prod of strides 2*5*5*6)
PERIOD = 2400        # lcm(HOP, FRAME): 5 hops = 8 frames
LOOKAHEAD = 1024     # n_fft / 2 of the prosody log-mel front-end


class _Edge:
    """(B, C, hist + max_new) buffer: [history | columns of the last push]; counts are host-side ints."""

    def __init__(self, sess, B, C, hist, max_new):
        self.hist = hist
        self.buf = torch.zeros(B, C, hist + max_new, device=sess.device, dtype=torch.float32)
        self.c = [0, 0]          # [columns pushed so far, columns of the last push]
        sess._counters.append(self)

    def push(self, x):
        n = x.shape[-1]
        ops.stream_push(self.buf, x, self.hist, self.c[1])
        self.c[0] += n
        self.c[1] = n

    def window(self, g0, length):
        col = self.hist + g0 - (self.c[0] - self.c[1])
        if col < 0 or col + length > self.hist + self.c[1]:
            raise RuntimeError(f"stream window [{g0}, {g0 + length}) outside the retained context")
        return self.buf[:, :, col:col + length]


class _Tap(_Edge):
    """Input side of one causal conv: history (k-1)*d, consumer cursor = outputs produced so far."""

    def __init__(self, sess, B, C, k, stride, dilation, max_new):
        super().__init__(sess, B, C, (k - 1) * dilation, max_new)
        self.k, self.s, self.d = k, stride, dilation
        self.pad = (k - 1) * dilation + 1 - stride
        self.c.append(0)         # c[2] = outputs so far

    def feed(self, x):
        """-> (x view, pad_left, pad_mode, t_out) for ops.conv1d."""
        first = self.c[0] == 0
        n = x.shape[-1]
        self.push(x)
        t_out = self.c[0] // self.s - self.c[2]
        if t_out <= 0:
            raise RuntimeError("streaming chunk too short to produce an output column")
        if first:
            if n % self.s or n <= self.pad:
                raise RuntimeError("first chunk must be a stride multiple longer than the reflect padding")
            view, pad_left, mode = self.window(0, n), self.pad, ops.PAD_REFLECT
        else:
            g0 = self.c[2] * self.s - self.pad
            view = self.window(g0, (t_out - 1) * self.s + (self.k - 1) * self.d + 1)
            pad_left, mode = 0, ops.PAD_ZERO
        self.c[2] += t_out
        return view, pad_left, mode, t_out


class _LSTMState:
    def __init__(self, sess, slstm, B):
        H, L = slstm.dimension, slstm.num_layers
        self.m, self.H = slstm, H
        self.state = [torch.zeros(3, H, ops.pad32(B), device=sess.device) for _ in range(L)]
        p = slstm.lstm
        self.w_ih = [ops.pack_conv_weight(getattr(p, f"weight_ih_l{l}").detach()) for l in range(L)]
        self.bias = [ops.add(getattr(p, f"bias_ih_l{l}").detach(), getattr(p, f"bias_hh_l{l}").detach()) for l in range(L)]
        self.whh = [ops.pack_lstm_whh(getattr(p, f"weight_hh_l{l}").detach()) for l in range(L)]
        self.c = [0]             # steps taken
        self._pre = {}           # (layer, T) -> zero-initialised (4H, T, BP) pre-activation buffer, real batch columns rewritten per hop
        sess._counters.append(self)

    def run(self, x, alpha_out):
        """SLSTM.forward (dac/model/encodec.py:282-288) continued from the carried state."""
        B, H, T = x.shape
        inp = ops.lstm_to_time_major(x)
        BP = inp.shape[2]
        few = LSTM_REAL_COLUMNS and T * B <= 4 and B < BP
        for l in range(len(self.state)):
            if few:
                # The recurrence kernel wants the batch padded to 32 columns; the input projection does not: as T "clips" of B
                # columns (views of the time-major buffers) it is a 1 - 4 column conv on the single-launch kernel instead of a
                # 32 T column one whose other columns are padding (33 -> 10 us per layer of the decoder's LSTM, tools/tune/hop_layers.py)
                pre = self._pre.get((l, T))
                if pre is None:
                    pre = self._pre[(l, T)] = torch.zeros(4 * H, T, BP, device=x.device)
                ops.conv1d(inp.view(H, T, BP).permute(1, 0, 2)[:, :, :B], self.w_ih[l], 4 * H, 1, bias=self.bias[l], pad_left=0,
                           t_out=B, pad_mode=ops.PAD_ZERO, out=pre.permute(1, 0, 2)[:, :, :B])
            else:
                pre = ops.conv1d(inp.view(1, H, T * BP), self.w_ih[l], 4 * H, 1, bias=self.bias[l], pad_left=0,
                                 t_out=T * BP, pad_mode=ops.PAD_ZERO).view(4 * H, T, BP)
            inp = ops.lstm_layer(pre.view(4 * H, T, BP), self.whh[l], H, state=self.state[l], step0=self.c[0])
        self.c[0] += T
        return ops.lstm_from_time_major(inp, x if self.m.skip else None, B, alpha_out)


def _conv(m, tap, x, **kw):
    view, pad_left, mode, t_out = tap.feed(x)
    w = m.w
    return ops.conv1d(view, w.packed(), w.c_out, m.kernel_size, bias=w.bias, stride=m.stride, dilation=m.dilation,
                      pad_left=pad_left, pad_mode=mode, t_out=t_out, **kw)


class _RUStream:
    """ResidualUnit (dac/model/dac.py:25-42) with a left-context buffer in front of its k7 conv."""

    def __init__(self, sess, ru, B, C, max_new):
        self.ru = ru
        self.tap = _Tap(sess, B, C, 7, 1, ru.block[1].dilation, max_new)

    def run(self, x, x_act, alpha_next, want_raw):
        b = self.ru.block
        k7, k1 = b[1], b[3]
        # small chunks: two split-reduction launches beat the fused single-tile kernel (latency, not throughput)
        if k7.w.c_out in FUSED_RU_CHANNELS and x.shape[0] * x.shape[-1] > 640:
            pair = _conv(k7, self.tap, x_act, alpha_out=b[2].flat(), res=x, w_k1=k1.w.packed(), bias_k1=k1.w.bias,
                         alpha_y2=alpha_next, want_y=want_raw or alpha_next is None)
            return pair if alpha_next is not None else (pair, None)
        h = _conv(k7, self.tap, x_act, alpha_out=b[2].flat())
        if alpha_next is None:
            return k1.run(h, res=x), None
        return k1.run(h, res=x, alpha_y2=alpha_next, want_y=want_raw)


def _run_units(units, x, x_act, alpha_after):
    for j, u in enumerate(units):
        last = j == len(units) - 1
        nxt = alpha_after if last else units[j + 1].ru.alpha_in
        x, x_act = u.run(x, x_act, alpha_next=nxt, want_raw=not last)
    return x_act


class _EncoderStream:
    """Encoder.forward (dac/model/dac.py:103-104) one chunk at a time; same fused plan as dac_model.Encoder."""

    def __init__(self, sess, enc, B, max_new):
        mods = list(enc.block)
        self.enc, self.mods = enc, mods
        self.blocks = [m for m in mods if isinstance(m, EncoderBlock)]
        self.tap0 = _Tap(sess, B, 1, 7, 1, 1, max_new)
        self.units, self.down = [], []
        rate = 1
        for blk in self.blocks:
            b = blk.block
            C = b[0].block[1].w.c_out
            self.units.append([_RUStream(sess, b[i], B, C, max_new // rate) for i in range(3)])
            self.down.append(_Tap(sess, B, C, b[4].kernel_size, b[4].stride, 1, max_new // rate))
            rate *= b[4].stride
        self.rate = rate
        self.lstm = _LSTMState(sess, mods[-3], B) if enc.use_lstm else None
        self.tap_out = _Tap(sess, B, enc.enc_dim, mods[-1].kernel_size, 1, 1, max_new // rate)

    def run(self, wave):
        mods, blocks = self.mods, self.blocks
        final_alpha = mods[-2].flat()
        x, x_act = _conv(mods[0], self.tap0, wave, alpha_y2=blocks[0].alpha_in)
        for i, blk in enumerate(blocks):
            b = blk.block
            if i + 1 < len(blocks):
                nxt = blocks[i + 1].alpha_in
            else:
                nxt = None if self.lstm is not None else final_alpha
            z_act = _run_units(self.units[i], x, x_act, b[3].flat())
            if nxt is not None:
                x, x_act = _conv(b[4], self.down[i], z_act, alpha_y2=nxt)
            else:
                x, x_act = _conv(b[4], self.down[i], z_act), None
        if self.lstm is not None:
            x_act = self.lstm.run(x, final_alpha)
        return _conv(mods[-1], self.tap_out, x_act)


class _DecoderStream:
    """Decoder.forward (dac/model/dac.py:164-165) one chunk of frames at a time."""

    def __init__(self, sess, dec, B, max_frames):
        mods = list(dec.model)
        self.dec, self.mods = dec, mods
        self.blocks = [m for m in mods if isinstance(m, DecoderBlock)]
        self.tap0 = _Tap(sess, B, mods[0].w.c_in, 7, 1, 1, max_frames)
        self.lstm = _LSTMState(sess, mods[1], B) if dec.use_lstm else None
        self.up, self.units = [], []
        n = max_frames
        for blk in self.blocks:
            b = blk.block
            self.up.append(_Tap(sess, B, b[1].w.c_in, 2, 1, 1, n))     # x[t-1] of the 2-tap polyphase form
            n *= b[1].stride
            C = b[1].w.c_out
            self.units.append([_RUStream(sess, b[i], B, C, n) for i in (2, 3, 4)])
        self.tap_out = _Tap(sess, B, mods[-2].w.c_in, 7, 1, 1, n)

    def _convtr(self, m, tap, x_act, alpha_y2):
        first = tap.c[0] == 0
        n = x_act.shape[-1]
        tap.push(x_act)
        tap.c[2] += n
        w = m.w
        if first:
            return ops.conv_transpose1d(tap.window(0, n), w.packed(), w.c_out, m.stride, bias=w.bias, alpha_y2=alpha_y2)
        return ops.conv_transpose1d(tap.window(tap.c[0] - n - 1, n + 1), w.packed(), w.c_out, m.stride, bias=w.bias,
                                    alpha_y2=alpha_y2, has_history=True)

    def run(self, z):
        mods, blocks = self.mods, self.blocks
        final_alpha = mods[-3].flat()
        if self.lstm is not None:
            x = _conv(mods[0], self.tap0, z)
            x_act = self.lstm.run(x, blocks[0].alpha_in)
        else:
            _, x_act = _conv(mods[0], self.tap0, z, alpha_y2=blocks[0].alpha_in, want_y=False)
        for i, blk in enumerate(blocks):
            b = blk.block
            nxt = blocks[i + 1].alpha_in if i + 1 < len(blocks) else final_alpha
            y, y_act = self._convtr(b[1], self.up[i], x_act, b[2].alpha_in)
            x_act = _run_units(self.units[i], y, y_act, nxt)
        return _conv(mods[-2], self.tap_out, x_act, act=ops.ACT_TANH)


class _QuantizerStream:
    """FAquantizer.forward_v2 (modules/quantize.py:375-454), eval, per chunk of frames, with a fixed timbre."""

    def __init__(self, sess, q, B, timbre, max_frames, wave_cap):
        self.q = q
        self.wave = _Edge(sess, B, 1, 2 * LOOKAHEAD, wave_cap)              # sample history for the STFT frames
        self.z_fifo = _Edge(sess, B, q.in_dim, 4, max_frames)               # latents waiting for look-ahead
        wn = q.melspec_encoder
        self.wn_taps = [_Tap(sess, B, wn.hidden_channels, 5, 1, 1, max_frames) for _ in range(wn.n_layers)]
        self.style = q.timbre_linear(timbre).contiguous()                   # (B, 2D) = [gamma | beta], fixed
        self.c = [0]                                                        # frames quantized so far
        sess._counters.append(self)
        self._rvq_w = {}

    def frames_ready(self, n_samples, final):
        return n_samples // FRAME if final else max(0, (n_samples - LOOKAHEAD) // FRAME + 1)

    def _mel(self, f0, n, final):
        fe = self.q.to_mel
        basis, fbp, off = fe._consts(self.wave.buf.device)
        total = self.wave.c[0]
        if f0 == 0:            # start of the signal: offline reflect padding on the left
            view, pad = self.wave.window(0, total), fe.n_fft // 2
        else:                  # interior (or, at finish(), reflect padding of the END of the signal)
            s0 = FRAME * f0 - fe.n_fft // 2
            length = total - s0 if final else FRAME * (n - 1) + fe.n_fft
            view, pad = self.wave.window(s0, length), 0
        w = view.reshape(view.shape[0], view.shape[-1])
        if not w.is_contiguous():
            w = w.contiguous()
        F_ = fe.n_fft // 2 + 1
        frames = ops.stft_frames(w, fe.win, n, fe.hop, pad, off)
        spec = ops.conv1d(frames, basis, 2 * F_, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=n)
        power = ops.spec_power(spec, 2)
        return ops.conv1d(power, fbp, fe.n_mels, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=n, act=ops.ACT_LOG_MEL)

    def _rvq(self, rvq, z, n_q):
        """ResidualVectorQuantize.forward (dac/nn/quantize.py:127-198), eval, codes + z_q only."""
        B, D, T = z.shape
        z_q = torch.zeros_like(z)
        codes = torch.empty(B, n_q, T, device=z.device, dtype=torch.int64)
        residual = torch.empty_like(z) if n_q > 1 else None
        src = z
        for i in range(n_q):
            vq = rvq.quantizers[i]
            if vq not in self._rvq_w:
                self._rvq_w[vq] = vq._weights()
            w_in, w_out, w_out_scale = self._rvq_w[vq]
            ops.vq_step(src, w_in, vq.in_proj.bias.detach(), vq.codebook.weight.detach(), w_out, w_out_scale,
                        vq.out_proj.bias.detach(), codes[:, i], residual=residual if i < n_q - 1 else None, zq_acc=z_q)
            src = residual
        return z_q, codes

    def push(self, wave_new, z_new):
        if wave_new is not None:
            self.wave.push(wave_new)
        if z_new is not None:
            self.z_fifo.push(z_new)

    def take_latents(self, final=False):
        """The latents of the frames whose look-ahead is complete, copied out of the FIFO -- or None when some of them are
        still to be produced by the encoder call of this very step (prime(), finish()).  In a steady-state hop they all come
        from earlier hops (the prosody front-end lags 1 024 samples, a hop brings 480), so the quantizer + decoder half of
        the hop does not depend on the encoder half and the two can run on two streams."""
        f0 = self.c[0]
        n = self.frames_ready(self.wave.c[0], final) - f0
        if n <= 0 or f0 + n > self.z_fifo.c[0]:
            return None
        return self.z_fifo.window(f0, n).contiguous()

    def prosody(self, final=False):
        """The branch of the frames whose look-ahead is complete that depends on the WAVEFORM only (log-mel -> 1x1 -> WaveNet -> 1x1 ->
        prosody RVQ, modules/quantize.py:398-413) -> (n, z_p, codes_p) or None."""
        q = self.q
        f0 = self.c[0]
        n = self.frames_ready(self.wave.c[0], final) - f0
        if n <= 0:
            return None
        mel = self._mel(f0, n, final)
        h = ops.conv1d(mel[:, :20], q.melspec_linear.w.packed(), 256, 1, bias=q.melspec_linear.w.bias, pad_left=0,
                       pad_mode=ops.PAD_ZERO, t_out=n)
        wn = q.melspec_encoder
        out = torch.zeros_like(h)
        fold = ops.STREAM_FOLD and h.shape[0] * n <= ops.SKINNY_MAX_COLS
        for i in range(wn.n_layers):                       # WN.forward, modules/wavenet.py:138-166
            last = i == wn.n_layers - 1
            if not fold:
                a = _conv(wn.in_layers[i], self.wn_taps[i], h)
                rs = wn.res_skip_layers[i].run(ops.gate_tanh_sigmoid(a))
                ops.wn_res_skip_(rs, h, out, last=last)
                continue
            # the same arithmetic with the gate and the residual / skip adds as epilogues of the two convs' reduction kernels
            acts = _conv(wn.in_layers[i], self.wn_taps[i], h, act=ops.ACT_GATE)
            rsl = wn.res_skip_layers[i]
            w = rsl.w
            if last:
                ops.conv1d(acts, w.packed(), w.c_out, 1, bias=w.bias, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=n, res=out, out=out)
            else:
                ops.conv1d(acts, w.packed(), w.c_out, 1, bias=w.bias, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=n, res=h, out=h,
                           skip_acc=out, act=ops.ACT_WN_RES_SKIP)
        f0_feat = q.melspec_linear2.run(out)
        z_p, codes_p = self._rvq(q.prosody_quantizer, f0_feat, 1)
        return n, z_p, codes_p

    def content(self, n_c, x):
        """Content RVQ of the due latents (modules/quantize.py:415-420): depends on the latents only."""
        return self._rvq(self.q.content_quantizer, x, n_c)

    def rest(self, x, pros, cont):
        """Residual RVQ of what the first two leave, sum, timbre-conditioned LayerNorm (modules/quantize.py:422-453)."""
        q = self.q
        n, z_p, codes_p = pros
        z_c, codes_c = cont
        z_r, codes_r = self._rvq(q.residual_quantizer, ops.sub2(x, z_p, z_c), 3)
        outs = ops.layernorm_c_affine(ops.add(ops.add(z_p, z_c), z_r), self.style)
        self.c[0] += n
        return outs, [codes_p, codes_c, codes_r]

    def run(self, n_c, final=False, x=None):
        """Quantizes the frames whose look-ahead is complete -> (outs, [codes_p, codes_c, codes_r]) or None.
        x: their latents if `take_latents` already copied them out."""
        f0 = self.c[0]
        pros = self.prosody(final)
        if pros is None:
            return None
        if x is None:
            x = self.z_fifo.window(f0, pros[0]).contiguous()
        return self.rest(x, pros, self.content(n_c, x))


class StreamingCodec:
    """One streaming session over B parallel streams.

        sess = StreamingCodec(model, timbre)            # model = build_model(...) (causal), timbre (B, 1024)
        out = sess.prime(wave[:, :, :4800])             # first chunk, >= 4 800 samples, multiple of 2 400
        out = sess.push(wave[:, :, t:t + 480])          # every hop: dict(codes=[p, c, r], wave=(B, 1, 300 n))
        out = sess.finish()                             # frames that were waiting for look-ahead

    Each call returns the frames completed by it (`frames` = index of the first one).  With graphs enabled the
    returned tensors are static buffers, valid until the next call.
    """

    def __init__(self, model, timbre, n_c=2, prime_samples=4800, use_graphs=True):
        enc, q, dec = model.encoder, model.quantizer, model.decoder
        if prime_samples % PERIOD or prime_samples < 2 * PERIOD:
            raise ValueError(f"prime_samples must be a multiple of {PERIOD} and at least {2 * PERIOD}")
        self.device = timbre.device
        self.B, self.n_c, self.prime_samples = timbre.shape[0], n_c, prime_samples
        self._counters = []
        for m in list(enc.modules()) + list(q.modules()) + list(dec.modules()):
            if isinstance(m, ConvWeights):
                m.freeze_packed = True               # inference: materialise w = g v/||v|| once
        B = self.B
        max_frames = prime_samples // FRAME
        self.enc = _EncoderStream(self, enc, B, prime_samples)
        self.qs = _QuantizerStream(self, q, B, timbre, max_frames, prime_samples)
        self.dec = _DecoderStream(self, dec, B, max_frames)
        self.n_samples = 0
        self.hops = 0
        self.use_graphs = use_graphs
        self.two_streams = os.environ.get("FAC_STREAM_TWO_STREAMS", "1") != "0"
        self._side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        if self._side is not None:
            ops.register_stream_slot(self._side)            # its own split-reduction scratch
        self._graphs = {}
        self._snap = {}
        self._hop_in = torch.zeros(B, 1, HOP, device=self.device)

    # ------------------------------------------------------------------------------------------ steps
    def _step(self, wave_new, final=False):
        first = self.qs.c[0]
        if wave_new is not None and self.two_streams and self._side is not None:
            # steady-state hop: [encoder -> latent FIFO] and [quantizer -> decoder] touch disjoint state once the samples are in
            # the STFT history and the due latents are copied out -- two chains of ~100 latency-bound launches side by side
            self.qs.push(wave_new, None)
            x = self.qs.take_latents(final)
            if x is not None:
                main = torch.cuda.current_stream(self.device)
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    outs, codes = self.qs.run(self.n_c, final, x=x)
                    wave = self.dec.run(outs)
                self.qs.push(None, self.enc.run(wave_new))
                # join: everything later on the caller's stream (reading the outputs, the next hop's fork) is ordered behind both
                main.wait_stream(self._side)
                return dict(frame0=first, codes=codes, wave=wave)
            self.qs.push(None, self.enc.run(wave_new))
        elif wave_new is not None:
            z = self.enc.run(wave_new)
            self.qs.push(wave_new, z)
        r = self.qs.run(self.n_c, final)
        if r is None:
            return dict(frame0=first, codes=None, wave=None)
        outs, codes = r
        return dict(frame0=first, codes=codes, wave=self.dec.run(outs))

    def prime(self, wave):
        if self.n_samples:
            raise RuntimeError("prime() must be the first call")
        if wave.shape[-1] != self.prime_samples:
            raise ValueError(f"prime() wants exactly {self.prime_samples} samples")
        self.n_samples = wave.shape[-1]
        return self._step(wave.contiguous())

    def _state(self):
        return [list(o.c) for o in self._counters]

    def _set_state(self, base, delta, k):
        for o, b, d in zip(self._counters, base, delta):
            o.c[:] = [bi + k * di for bi, di in zip(b, d)]

    def push(self, hop):
        if not self.n_samples:
            raise RuntimeError("call prime() first")
        if hop.shape[-1] != HOP:
            raise ValueError(f"push() wants exactly {HOP} samples")
        h, phase = self.hops, self.hops % 5
        self.hops += 1
        self.n_samples += HOP
        if not self.use_graphs:
            return self._step(hop.contiguous())
        self._hop_in.copy_(hop)
        if h < 5:                                         # first period: eager (also warms every kernel up)
            out = self._step(self._hop_in)
            self._snap[h] = self._state()
            return out
        if h < 10:                                        # second period: capture one graph per phase
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._step(self._hop_in)
            g.replay()                                    # capture does not execute
            after = self._state()
            delta = [[a - b for a, b in zip(sa, sb)] for sa, sb in zip(after, self._snap[h - 5])]
            self._graphs[phase] = (g, out, after, delta, h)
            return out
        g, out, base, delta, h0 = self._graphs[phase]
        g.replay()
        k = (h - h0) // 5
        self._set_state(base, delta, k)
        return dict(out, frame0=out["frame0"] + 8 * k)

    def finish(self):
        """End of the signal (length must be a multiple of 300): emits the remaining frames."""
        if self.n_samples % FRAME:
            raise ValueError("finish(): stream length must be a multiple of 300 samples")
        return self._step(None, final=True)
