"""``Config`` of the reference (src/v2/utils.py:25-43): same 15 fields, defaults and ``__str__``.

One extra field, ``generator_kind`` ("the working tail behind an explicit Config extra field"): it is
excluded from ``repr`` so ``str(Config())`` stays byte-identical to the reference's, and it defaults to the reference
behaviour."""
from pydantic import BaseModel, Field

GENERATOR_KINDS = ("v2", "sln_siren", "sln_siren_patch")


class Config(BaseModel):
    attention_heads_count: int = 4
    batch_size: int = 64
    classes_count: int = 10
    discriminator_learning_rate: float = 5e-4
    dropout_rate: float = 0.1
    embeddings_dimension: int = 128
    epochs: int = 500
    generator_learning_rate: float = 5e-4
    image_size: int = 32
    input_channels: int = 3
    mlp_ratio: int = 2
    optimizer_beta1: float = 0.5
    optimizer_beta2: float = 0.999
    patch_size: int = 4
    transformer_blocks_count: int = 6
    # what ``ViTGAN(config).generator`` is:
    #   "v2"              the reference's ViTGenerator (src/v2/modules.py:344-372), whose tail cannot produce an image;
    #   "sln_siren"       the v1 SLN/SIREN generator (src/v1/generator.py:12-69), one token per image row - a working G;
    #   "sln_siren_patch" the same blocks on the discriminator's patch grid (for images beyond 32x32).
    generator_kind: str = Field(default="v2", repr=False)

    def __str__(self):
        body = repr(self)
        return "\n".join(body[body.index("(") + 1: -1].split(", "))
