"""Model factory with the reference's surface (modules/commons.py:283-348, :446-479):

    model = build_model(recursive_munch(config['model_params']))      # stage='codec'
    z = model.encoder(wave); z, quantized, commit, codebook, timbre = model.quantizer(z, wave, n_c=2)
    wave_hat = model.decoder(z)

so reconstruct.py / train.py-style drivers keep working unchanged.  `munch` is not installed in this
image, hence the small attribute-dict below (same behaviour for the keys the callers use).
"""
import torch


class Munch(dict):
    """dict with attribute access (stand-in for munch.Munch, modules/commons.py:6)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def recursive_munch(d):
    """modules/commons.py:473-479."""
    if isinstance(d, dict):
        return Munch((k, recursive_munch(v)) for k, v in d.items())
    if isinstance(d, list):
        return [recursive_munch(v) for v in d]
    return d


def build_model(args, stage="codec"):
    """modules/commons.py:283-348.  Returns Munch(encoder, quantizer, decoder, discriminator, fa_predictors)
    with the reference's key order."""
    from .dac_model import Encoder, Decoder
    from .quantize import FAquantizer

    if stage == "redecoder":      # modules/commons.py:385-413 (discriminator: train-only, not built)
        from .redecoder import Redecoder
        return Munch(encoder=Redecoder(args),
                     decoder=Decoder(input_channel=1024, channels=args.DAC.decoder_dim, rates=args.DAC.decoder_rates,
                                     causal=args.decoder_causal, lstm=args.decoder_lstm))
    if stage == "encoder":        # modules/commons.py:414-439
        return Munch(encoder=Encoder(d_model=args.DAC.encoder_dim, strides=args.DAC.encoder_rates, d_latent=1024,
                                     causal=args.encoder_causal, lstm=args.encoder_lstm),
                     quantizer=FAquantizer(in_dim=1024, n_p_codebooks=1, n_c_codebooks=args.n_c_codebooks,
                                           n_t_codebooks=2, n_r_codebooks=3, codebook_size=1024, codebook_dim=8,
                                           quantizer_dropout=0.5, causal=args.encoder_causal,
                                           separate_prosody_encoder=args.separate_prosody_encoder,
                                           timbre_norm=args.timbre_norm))
    if stage != "codec":
        raise ValueError(f"Unknown stage: {stage}")

    encoder = Encoder(d_model=args.DAC.encoder_dim, strides=args.DAC.encoder_rates, d_latent=1024,
                      causal=args.causal, lstm=args.lstm)
    quantizer = FAquantizer(in_dim=1024, n_p_codebooks=1, n_c_codebooks=args.n_c_codebooks, n_t_codebooks=2,
                            n_r_codebooks=3, codebook_size=1024, codebook_dim=8, quantizer_dropout=0.5,
                            causal=args.causal, separate_prosody_encoder=args.separate_prosody_encoder,
                            timbre_norm=args.timbre_norm)
    decoder = Decoder(input_channel=1024, channels=args.DAC.decoder_dim, rates=args.DAC.decoder_rates,
                      causal=args.causal, lstm=args.lstm)
    from .predictors import FApredictors
    fa_predictors = FApredictors(in_dim=1024, use_gr_content_f0=args.use_gr_content_f0,
                                 use_gr_prosody_phone=args.use_gr_prosody_phone, use_gr_residual_f0=True,
                                 use_gr_residual_phone=True, use_gr_timbre_content=True,
                                 use_gr_timbre_prosody=args.use_gr_timbre_prosody, use_gr_x_timbre=True,
                                 norm_f0=args.norm_f0, timbre_norm=args.timbre_norm,
                                 use_gr_content_global_f0=args.use_gr_content_global_f0)
    from .discriminator import Discriminator
    discriminator = Discriminator(rates=[], periods=[2, 3, 5, 7, 11], fft_sizes=[2048, 1024, 512],
                                  sample_rate=args.DAC.sr)                      # modules/commons.py:334-340
    return Munch(encoder=encoder, quantizer=quantizer, decoder=decoder, discriminator=discriminator,
                 fa_predictors=fa_predictors)


def default_model_params():
    """configs/config.yml:27-46 `model_params`."""
    return recursive_munch(dict(
        fixed=True, causal=True, lstm=2, norm_f0=True, use_gr_content_f0=False, use_gr_prosody_phone=False,
        use_gr_timbre_prosody=False, separate_prosody_encoder=True, n_c_codebooks=2, timbre_norm=True,
        use_gr_content_global_f0=True, w2v="w2v-ctc",
        DAC=dict(encoder_dim=64, encoder_rates=[2, 5, 5, 6], decoder_dim=1536, decoder_rates=[6, 5, 5, 2], sr=24000)))


def default_redecoder_params():
    """configs/config_redecoder.yml:28-48 `model_params`."""
    return recursive_munch(dict(
        encoder_causal=True, decoder_causal=False, encoder_lstm=2, decoder_lstm=0, n_c_codebooks=2, n_p_codebooks=1,
        timbre_norm=True, separate_prosody_encoder=True, encoder_type="wavenet", wavenet_embed_dim=512,
        mamba_embed_dim=768, prob_random_mask_prosody=1.0, prob_random_mask_content=[0.0, 1.0],
        DAC=dict(encoder_dim=64, encoder_rates=[2, 5, 5, 6], decoder_dim=1536, decoder_rates=[6, 5, 5, 2], sr=24000)))


def load_checkpoint(model, optimizer, path, load_only_params=True, ignore_modules=(), is_distributed=False):
    """modules/commons.py:446-471: {'net': {key: state_dict}, ...}; strips DDP's 'module.' prefix."""
    state = torch.load(path, map_location="cpu")
    params = state["net"]
    for key in model:
        if key in params and key not in ignore_modules:
            sd = params[key]
            if not is_distributed:
                sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
            model[key].load_state_dict(sd, strict=True)
    for key in model:
        model[key].eval()
    epoch, iters = state.get("epoch", 0) + 1, state.get("iters", 0)
    if not load_only_params and optimizer is not None:
        optimizer.load_state_dict(state["optimizer"])
        optimizer.load_scheduler_state_dict(state["scheduler"])
    return model, optimizer, epoch, iters
