"""Voice-conversion re-decoder (reference: modules/redecoder.py:4-48, `encoder_type: wavenet`; call site
reconstruct_redecoder.py:110-122): code embeddings summed per frame -> 16-layer WaveNet conditioned on the
timbre vector -> 1x1 conv back to the 1024-d latent that the (non-causal, LSTM-free) Decoder consumes.
The `mamba` branch of the reference imports a module that is not in its tree (modules.mamba): not built."""
import torch
from torch import nn

from . import ops
from .quantize import WN, _PlainConv


class _Embedding(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n, d))


class Redecoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.n_p_codebooks, self.n_c_codebooks = args.n_p_codebooks, args.n_c_codebooks
        self.codebook_size = 1024
        self.encoder_type = args.encoder_type
        if args.encoder_type != "wavenet":
            raise NotImplementedError("only encoder_type 'wavenet' is built (the reference's 'mamba' module is absent)")
        self.embed_dim = args.wavenet_embed_dim
        self.encoder = WN(hidden_channels=self.embed_dim, kernel_size=5, dilation_rate=1, n_layers=16, gin_channels=1024,
                          p_dropout=0.2, causal=args.decoder_causal)
        self.conv_out = _PlainConv(self.embed_dim, 1024, 1)
        self.prosody_embed = nn.ModuleList([_Embedding(self.codebook_size, self.embed_dim) for _ in range(self.n_p_codebooks)])
        self.content_embed = nn.ModuleList([_Embedding(self.codebook_size, self.embed_dim) for _ in range(self.n_c_codebooks)])

    def forward(self, p_code, c_code, timbre_vec, use_p_code=True, use_c_code=True, n_c=2):
        """p_code (B, n_p, T), c_code (B, >= n_c, T) int64; timbre_vec (B, 1024) -> (B, 1024, T)."""
        B, _, T = p_code.shape
        x = torch.zeros(B, self.embed_dim, T, device=p_code.device, dtype=torch.float32)
        if use_p_code and self.n_p_codebooks:
            tabs = torch.stack([e.weight.detach() for e in self.prosody_embed])
            ops.embed_sum(p_code, tabs, 0, out=x)
        if use_c_code and n_c:
            tabs = torch.stack([e.weight.detach() for e in list(self.content_embed)[:n_c]])
            ops.embed_sum(c_code, tabs, 0, out=x)
        x = self.encoder(x, None, g=timbre_vec)
        return self.conv_out.run(x)
