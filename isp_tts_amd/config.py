"""Model dimensions of the acoustic model, as plain dataclasses.

The defaults are the reference recipe `recipes/acoustic/core.yaml:85-158` (encoder/decoder
dim 384 x depth 6, 6 heads x 64, one shared K/V head, ALiBi, FFN 1536 GELU; adaptor predictor
dim 256 x depth 3 with a 32-d time embedding; embedding transformer dim 256 x depth 1;
aligner attention_dim 128, kernels 5 / [5, 5]; mel_dim 80).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict


def _layer(heads: int, inner: int, dropout: float) -> dict:
    return {
        "attention": {"heads": heads, "head_dim": 64, "dropout": dropout,
                      "one_kv_head": True, "alibi_pos_bias": True},
        "feed_forward": {"inner_dim": inner, "dropout": dropout, "activation": "gelu"},
        "pre_norm": True,
    }


@dataclass
class AcousticDims:
    vocab: int = 149            # en_ipa symbol table (V=149; ru_en is 77)
    mel_dim: int = 80
    text_dim: int = 384
    enc_depth: int = 6
    dec_depth: int = 6
    heads: int = 6
    ffn: int = 1536
    ada_dim: int = 256
    ada_depth: int = 3
    ada_heads: int = 4
    ada_ffn: int = 1024
    emb_depth: int = 1
    time_dim: int = 32
    attn_dim: int = 128         # aligner attention_dim
    key_kernel: int = 5
    query_kernels: tuple = (5, 5)

    def model_config(self) -> dict:
        """Nested dict in the shape `AcousticModel.init(config)` takes (model.py:60-72 of the reference)."""
        enc = {"dim": self.text_dim, "depth": self.enc_depth, "transformer_layer": _layer(self.heads, self.ffn, 0.1)}
        dec = {"dim": self.text_dim, "depth": self.dec_depth, "transformer_layer": _layer(self.heads, self.ffn, 0.1)}
        ada_layer = _layer(self.ada_heads, self.ada_ffn, 0.3)
        return {
            "encoding_map": {str(i): i for i in range(self.vocab)},
            "mel_dim": self.mel_dim,
            "text_dim": self.text_dim,
            "encoder": enc,
            "decoder": dec,
            "temporal_adaptor": {
                "predictor": {"time_embedding_dim": self.time_dim,
                              "transformer": {"dim": self.ada_dim, "depth": self.ada_depth,
                                              "transformer_layer": ada_layer}},
                "embedding": {"transformer": {"dim": self.ada_dim, "depth": self.emb_depth,
                                              "transformer_layer": ada_layer}},
                "pitch": True, "energy": True, "soft_duration": True,
            },
            "aligner": {"attention_dim": self.attn_dim, "key_kernel_size": self.key_kernel,
                        "query_kernel_size": list(self.query_kernels), "dropout": 0.1,
                        "normalization": "instance", "activation": "gelu"},
            "num_speakers": None,
        }
