"""The training iteration of the reference (train.py:188-374) on the HIP path.

    wav_seg, ... = crop_segments(waves, mel_lengths, ...)                                         train.py:188-212
    z = encoder(wav_seg);  outs, quantized, commitment, codebook, timbre = quantizer(z, wav_seg, n_c=2,
                                                                   full_waves=waves, wave_lens=wave_lengths)  :265-269
    preds, rev_preds = fa_predictors(quantized, timbre);  pred = decoder(outs)                    :270-277
    discriminator:  loss_d = sum_k mean(D_k(pred.detach())^2) + mean((1 - D_k(target))^2); clip 10; AdamW    :279-292
    generator:      15 mel + feature matching + adversarial + 0.25 commitment + codebook
                    + f0 + uv + 5 focal(content) + speaker;  clip 1000;  AdamW x4                 :294-374

Data parallel: one asynchronous all-reduce(mean) of each key's gradient arena, launched as soon as that key's
gradients are final (decoder: from a hook on the decoder input's gradient, i.e. under the quantizer / encoder
backward; discriminator: under the mel-loss forward of the generator half), waited for in the key's optimiser step.
"""
import os

import torch

from . import losses, optim


def crop_segments(waves, mel_input_length, max_frame_len=80, hop=300, starts=None, generator=None, extra=()):
    """train.py:188-212 on the device.  waves (B, T_full) padded batch (device); mel_input_length: per-clip frame counts as a
    HOST sequence (the dataloader hands them over on the CPU, train.py:176 -- reading them costs no device sync);
    starts: optional (B,) int64 frame offsets (the reference draws np.random.randint(0, mel_length - seg_len) per clip,
    :195); extra: further (B, C, F) frame-rate tensors (mel targets, phone ids as float) cropped with the same offsets.
    Returns wav_seg (B, 1, seg_len * hop), starts (device int64), [cropped extras]."""
    from . import autograd_disc as AD
    lens = [int(v) for v in mel_input_length]
    seg = min(min(lens), int(max_frame_len))
    B = waves.shape[0]
    if starts is None:
        span = torch.tensor([max(n - seg, 0) for n in lens], dtype=torch.float64)
        u = torch.rand(B, generator=generator, dtype=torch.float64)
        starts = torch.minimum((u * span).floor(), (span - 1).clamp_min(0)).to(torch.int64)   # randint(0, len - seg); 0 when equal
    from . import ops
    starts = ops.h2d(torch.as_tensor(starts, dtype=torch.int64), waves.device)
    wav_seg = AD.CropRows.apply(waves.reshape(B, 1, -1), starts, seg * hop, hop)
    out_extra = [AD.CropRows.apply(e if e.dim() == 3 else e.reshape(B, 1, -1), starts, seg, 1).reshape(*e.shape[:-1], seg) for e in extra]
    return wav_seg, starts, out_extra


UNBIND_GRADS = os.environ.get("FAC_UNBIND_GRADS", "1") != "0"


class GeneratorStep:
    """Generator half without the GAN / predictor terms: 15 mel + 0.25 commitment + codebook (train.py:357-358 subset)."""

    def __init__(self, model, lr=1e-4, sample_rate=24000, broadcast=True):
        """broadcast: with a process group of more than one rank, every rank starts from rank 0's parameters (what
        DistributedDataParallel's constructor does for train.py:110-111) -- one broadcast of each key's parameter arena."""
        self.model = model
        self._broadcast = broadcast
        for k in ("encoder", "quantizer", "decoder"):
            model[k].train()
        self.mel = losses.MelSpectrogramLoss(n_mels=[5, 10, 20, 40, 80, 160, 320], window_lengths=[32, 64, 128, 256, 512, 1024, 2048],
                                             mel_fmin=[0] * 7, mel_fmax=[None] * 7, pow=1.0, mag_weight=0.0, clamp_eps=1e-5,
                                             sample_rate=sample_rate)          # train.py:155-163
        self.opt = {"encoder": optim.FlatAdamW(model.encoder.parameters(), lr=lr),
                    "quantizer": optim.FlatAdamW(model.quantizer.parameters(), lr=lr),
                    "decoder": optim.FlatAdamW(model.decoder.parameters(), lr=lr)}
        self._sync_start(self.opt.values())
        self._install_boundary_hooks()

    def _sync_start(self, opts):
        if self._broadcast:
            for o in opts:
                o.broadcast_parameters(0)

    def _early_exchange(self, outs, z):
        """Launch the gradient exchange from inside backward, at points of the graph where the parameters concerned are final BY
        CONSTRUCTION -- the same points on every rank, whatever each rank's batch did (VERDICT r3 item 7 / ADVICE r3):

        * autograd runs ready nodes latest-created first and AccumulateGrad before anything else, so when the gradient of a
          tensor is handed on, every node created after that tensor's producer has run.  Hence: at the hook on the decoder input
          `outs` every decoder parameter is final; at the hook on the encoder output `z` the predictor heads and the whole
          quantizer are (side branches included: they were created after the encoder).  Those keys launch ALL their buckets there.
        * inside the encoder / decoder chains (`boundary_hook`, dac_model.py): when the gradient of the activation entering
          top-level child k is handed on, child k and everything behind it is done -- including the first Snake of child k + 1,
          whose alpha receives its gradient from a node of child k (the fused pre-activated copy).  So the buckets made only of
          parameters of children k + 1 .. end are launched at that hook (`launch_all_reduce(from_param=...)`): the decoder's
          342 MB and the encoder's 145 MB leave in <= 64 MB pieces while the rest of their backward still runs, and what is
          outstanding when backward ends is the encoder's first bucket only.
        A parameter a rank does not reach contributes zeros (and a zero flag), like DistributedDataParallel's unused parameters
        (train.py:49 `find_unused_parameters=True`); a gradient that arrives AFTER its bucket left raises in `FlatAdamW.step`.
        FAC_EARLY_EXCHANGE=0: everything is launched after backward, in one fixed order."""
        if os.environ.get("FAC_EARLY_EXCHANGE", "1") == "0":
            return
        opt = self.opt
        outs.register_hook(lambda g: opt["decoder"].launch_all_reduce(from_hook=True))

        def at_z(g):
            for k in ("fa_predictors", "quantizer"):
                if k in opt and (k != "fa_predictors" or getattr(self, "with_predictors", False)):
                    opt[k].launch_all_reduce(from_hook=True)
        z.register_hook(at_z)

    def _install_boundary_hooks(self):
        """Encoder / decoder: at the activation entering top-level child k, launch the buckets of children k + 1 .. end."""
        early = os.environ.get("FAC_EARLY_EXCHANGE", "1") != "0"
        for key, attr in (("encoder", "block"), ("decoder", "model")):
            o = self.opt[key]
            mod = getattr(self.model[key], "module", self.model[key])       # a DistributedDataParallel wrapper keeps it in .module
            if not early or not o.data_parallel:
                mod.boundary_hook = None
                continue
            seq = getattr(mod, attr)
            index = {id(p): i for i, p in enumerate(o.params)}
            children = list(seq)
            first = []                                     # first parameter index of every top-level child (None: no parameters)
            for m in children:
                idx = [index[id(p)] for p in m.parameters() if id(p) in index]
                first.append(min(idx) if idx else None)
            after = {}                                     # child -> first parameter index of the children behind it
            for k, m in enumerate(children):
                nxt = [f for f in first[k + 1:] if f is not None]
                after[id(m)] = min(nxt) if nxt else None

            def hook(m, x, o=o, after=after):
                fp = after.get(id(m))
                if fp is not None and x.requires_grad:
                    x.register_hook(lambda g, o=o, fp=fp: o.launch_all_reduce(from_param=fp, from_hook=True))
            mod.boundary_hook = hook

    def exchange_report(self):
        """Per key: where the last step's gradient exchange was launched from -- "hook" (inside backward: it overlaps the
        rest of the backward pass), "end" (after backward: no overlap), "none" (single rank) -- and, with
        `opt[k].time_exchange = True`, how long the compute stream stalled for it."""
        def summary(o):
            log = o.exchange_log[-1] if o.exchange_log else []
            if not log:
                return "none"
            return "hook" if log[0][1] == "hook" else "end"

        return {k: dict(launched=summary(o), buckets=(o.exchange_log[-1] if o.exchange_log else []), n_buckets=len(o.buckets),
                        wait_ms=o.exchange_wait_ms()) for k, o in self.opt.items()}

    _lstm_timeouts_seen = 0

    def _report_lstm_timeouts(self):
        """Host-side report of resident-LSTM waits that gave up (ops.lstm_timeouts: no synchronisation, so it speaks about steps the
        device has finished).  The affected step was skipped on every rank by the optimiser's poison word (optim.FlatAdamW) and the
        LSTM layers run on the per-step kernels from then on; this only makes sure it does not go unnoticed."""
        from . import ops
        n = ops.lstm_timeouts() if next(iter(self.opt.values())).p.is_cuda else 0
        if n > GeneratorStep._lstm_timeouts_seen:
            import warnings
            warnings.warn(f"facodec_amd: {n} resident-LSTM wait(s) timed out on this device so far; the step they fell into was skipped "
                          "by every rank's optimiser (no parameter moved) and the LSTM layers now use the per-step kernels")
            GeneratorStep._lstm_timeouts_seen = n

    def _zero(self, keys):
        """Gradient arenas cleared; `.grad` left unbound so that autograd hands over its gradient tensors and the
        optimiser folds them in with one multi-tensor copy (FlatAdamW.zero_grad).  Under a DistributedDataParallel
        wrapper (data_parallel=False) the reducer owns `.grad`: the views stay bound."""
        for k in keys:
            self.opt[k].zero_grad(unbind=UNBIND_GRADS and self.opt[k].data_parallel)

    def forward_backward(self, wave, masks=None, full_waves=None, wave_lens=None):
        m = self.model
        self._report_lstm_timeouts()
        self._zero(("encoder", "quantizer", "decoder"))
        z = m.encoder(wave)
        outs, _, commitment, codebook, _ = m.quantizer(z, wave, n_c=2, masks=masks, full_waves=full_waves, wave_lens=wave_lens)
        wave_hat = m.decoder(outs)
        mel = self.mel(wave_hat, wave)
        loss = 15.0 * mel + 0.25 * commitment + 1.0 * codebook
        self._early_exchange(outs, z)
        loss.backward()
        return dict(loss=loss.detach(), mel=mel.detach(), commitment=commitment.detach(), codebook=codebook.detach())

    def _weight_caches(self):
        """wprep.WeightCache per side: the weight-norm scales and packed layouts of every conv re-materialised as a few launches at
        the start of the step (and, for the discriminator, again after its optimiser step) instead of ~1 700 small ones inside it."""
        wc = self.__dict__.get("_wc")
        if wc is None:
            from . import wprep
            wc = self._wc = {}
            if wprep.ENABLED and next(iter(self.opt.values())).p.is_cuda:
                gen = [p for k, o in self.opt.items() if k != "discriminator" for p in o.params]
                wc["gen"] = wprep.WeightCache(gen, "generator side", any_thread=True)
                if "discriminator" in self.opt:
                    wc["disc"] = wprep.WeightCache(list(self.opt["discriminator"].params), "discriminator", any_thread=True)
        return wc

    def _in_weight_regions(self, fn, *a, **kw):
        caches = list(self._weight_caches().values())
        for c in caches:
            c.begin()
        try:
            return fn(*a, **kw)
        finally:
            for c in caches:
                c.end()

    def __call__(self, wave, masks=None, full_waves=None, wave_lens=None):
        return self._in_weight_regions(self._step, wave, masks, full_waves, wave_lens)

    def _step(self, wave, masks=None, full_waves=None, wave_lens=None):
        out = self.forward_backward(wave, masks, full_waves, wave_lens)
        for k in ("decoder", "quantizer", "encoder"):
            self.opt[k].launch_all_reduce()
        for k in ("encoder", "decoder", "quantizer"):                    # train.py:362-374
            self.opt[k].step(zero_grad=False)
        out["grad_norm"] = {k: self.opt[k].grad_norm() for k in self.opt}
        return out


class TrainStep(GeneratorStep):
    """Both halves of train.py's iteration (:265-374).  The predictor-head losses (:314-356) are included when
    `with_predictors=True` and the caller supplies their targets (the reference obtains them from a pitch extractor, a
    CTC phoneme model and a speaker model, none of which is part of this build).  After a call every `p.grad` still holds
    the (rank-averaged, unclipped) gradient of the step; the arenas are cleared at the start of the next call."""

    def __init__(self, model, lr=1e-4, sample_rate=24000, with_predictors=False, broadcast=True):
        super().__init__(model, lr, sample_rate, broadcast)
        model.discriminator.train()
        self.opt["discriminator"] = optim.FlatAdamW(model.discriminator.parameters(), lr=lr, max_norm=10.0)
        self.with_predictors = with_predictors
        if with_predictors:
            model.fa_predictors.train()
            self.opt["fa_predictors"] = optim.FlatAdamW(model.fa_predictors.parameters(), lr=lr)
        self._sync_start([self.opt[k] for k in ("discriminator", "fa_predictors") if k in self.opt])
        self.stft = losses.MultiScaleSTFTLoss()
        self.l1 = losses.L1Loss()
        self.batched_d_step = os.environ.get("FAC_BATCHED_D_STEP", "1") != "0"

    def predictor_losses(self, preds, rev, targets):
        """train.py:314-356 given the targets the reference takes from external models: f0 (B, F) normalised log-F0
        (-10 = unvoiced), uv (B, F) the log-normalised mel energy `real_norm`, phones (B, F) int64 in [0, 1024),
        speaker (B,) int64 in [0, 20000).  Returns (1*f0 + 1*uv + 5*content + 1*speaker (:357-358), the eight terms)."""
        from . import autograd_disc as AD
        n = min(preds["f0"].shape[-2], targets["f0"].shape[-1])
        f0_t, uv_t = targets["f0"][..., :n].contiguous(), targets["uv"][..., :n].contiguous()
        ph_t = targets["phones"][..., :n].contiguous().reshape(-1)

        def sl1(p, t):        # F.smooth_l1_loss(target, pred.squeeze(-1)[..., :n])
            return AD.PairMean.apply(p.squeeze(-1)[..., :n].contiguous(), t, 3)

        def focal(logits_btc):   # FocalLoss(gamma=2)(pred.transpose(1, 2)[..., :n], target) (:153, :334-336): rows (b, t)
            lg = logits_btc[:, :n].contiguous()
            return AD.focal_cross_entropy(lg.reshape(-1, lg.shape[-1]), ph_t, 2.0)

        t = dict(f0_loss=sl1(preds["f0"], f0_t), uv_loss=sl1(preds["uv"], uv_t), content_loss=focal(preds["content"]),
                 spk_loss=AD.CrossEntropy.apply(preds["timbre"], targets["speaker"]))
        if rev["rev_f0"] is not None:
            t["rev_f0_loss"] = sl1(rev["rev_f0"], f0_t)
        if rev["rev_uv"] is not None:
            t["rev_uv_loss"] = sl1(rev["rev_uv"], uv_t)
        if rev["rev_content"] is not None:
            t["rev_content_loss"] = focal(rev["rev_content"])
        if rev["x_timbre"] is not None:
            t["x_spk_loss"] = AD.CrossEntropy.apply(rev["x_timbre"], targets["speaker"])
        z = lambda k: t.get(k, 0.0)    # noqa: E731
        total = 1.0 * (t["f0_loss"] + z("rev_f0_loss")) + 1.0 * (t["uv_loss"] + z("rev_uv_loss")) \
            + 5.0 * (t["content_loss"] + z("rev_content_loss")) + 1.0 * (t["spk_loss"] + z("x_spk_loss"))
        return total, t

    def __call__(self, wave, masks=None, targets=None, full_waves=None, wave_lens=None, log_losses=False):
        return self._in_weight_regions(self._step, wave, masks, targets, full_waves, wave_lens, log_losses)

    def _step(self, wave, masks=None, targets=None, full_waves=None, wave_lens=None, log_losses=False):
        """wave (B, 1, T) the cropped segments; full_waves (B, T_full) / wave_lens (B,) the whole utterances for the timbre
        encoder (train.py:266-269); targets: predictor targets (with_predictors); log_losses: also evaluate the two
        logged-only criteria of train.py:296,298 (multi-scale STFT, waveform L1)."""
        from .discriminator import gan_losses
        m, opt = self.model, self.opt
        self._report_lstm_timeouts()
        self._zero(opt.keys())
        z = m.encoder(wave)
        outs, quantized, commitment, codebook, timbre = m.quantizer(z, wave, n_c=2, masks=masks, full_waves=full_waves,
                                                                    wave_lens=wave_lens)
        if self.with_predictors:
            if targets is None:
                raise ValueError("with_predictors=True needs targets (f0, uv, phones, speaker)")
            preds, rev = m.fa_predictors(quantized, timbre)
        pred = m.decoder(outs)
        target = wave
        len_diff = target.size(-1) - pred.size(-1)                       # train.py:274-276
        if len_diff > 0:
            target = target[..., len_diff // 2:-len_diff // 2].contiguous()
        # ---- discriminator step (train.py:279-292)
        disc = m.discriminator
        if self.batched_d_step:      # one pass over [fake | real]: half the launches, twice the tiles per launch
            from .discriminator import gan_loss_d_batched
            loss_d = gan_loss_d_batched(disc.forward_internal(torch.cat([pred.detach(), target], 0)))
        else:
            d_fake, d_real = disc.forward_internal(pred.detach()), disc.forward_internal(target)
            loss_d, _, _ = gan_losses(d_fake, d_real)
        loss_d.backward()
        opt["discriminator"].launch_all_reduce()                         # rides under the loss forwards below
        mel = self.mel(pred, target)
        extra = {}
        if log_losses:
            with torch.no_grad():
                extra = dict(stft=self.stft(pred, target), waveform=self.l1(pred, target))
        if self.with_predictors:
            pred_total, terms = self.predictor_losses(preds, rev, targets)
            extra.update({k: v.detach() for k, v in terms.items()})
        opt["discriminator"].step(zero_grad=False)
        wc = self._weight_caches().get("disc")
        if wc is not None and wc.depth == 1:                  # the generator step sees the UPDATED discriminator (train.py:292-300)
            wc.end()
            wc.begin()
        # ---- generator step (:295-374): the discriminator is only differentiated w.r.t. its input
        for p in opt["discriminator"].params:
            p.requires_grad_(False)
        try:
            d_fake = disc.forward_internal(pred)
            with torch.no_grad():
                d_real = disc.forward_internal(target)
            _, loss_g, loss_feat = gan_losses(d_fake, d_real)
            loss = 15.0 * mel + 1.0 * loss_feat + 1.0 * loss_g + 0.25 * commitment + 1.0 * codebook
            if self.with_predictors:
                loss = loss + pred_total
            self._early_exchange(outs, z)
            loss.backward()
        finally:
            for p in opt["discriminator"].params:
                p.requires_grad_(True)
        gen_keys = ("decoder", "quantizer", "encoder") + (("fa_predictors",) if self.with_predictors else ())
        for k in gen_keys:
            opt[k].launch_all_reduce()
        for k in ("encoder", "decoder", "quantizer") + (("fa_predictors",) if self.with_predictors else ()):
            opt[k].step(zero_grad=False)
        return dict(loss=loss.detach(), loss_d=loss_d.detach(), loss_g=loss_g.detach(), feature=loss_feat.detach(),
                    mel=mel.detach(), commitment=commitment.detach(), codebook=codebook.detach(),
                    grad_norm={k: opt[k].grad_norm() for k in opt}, **extra)
