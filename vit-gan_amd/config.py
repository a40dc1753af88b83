"""``Config`` of the reference (src/v2/utils.py:25-43): same 15 fields, defaults and ``__str__``."""
from pydantic import BaseModel


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

    def __str__(self):
        body = repr(self)
        return "\n".join(body[body.index("(") + 1: -1].split(", "))
