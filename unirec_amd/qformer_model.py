"""Drop-in for ``models/qformer_model.py:6-50``: the same class as qformer_utils' with the
reference's different default ``num_query_tokens=8``."""
from .qformer import BertConfig, BertModel  # noqa: F401  (re-exported like the reference module)
from .qformer_utils import QFormerForItemRepresentation as _Base


class QFormerForItemRepresentation(_Base):
    def __init__(self, hidden_size: int = 1024, num_hidden_layers: int = 12, num_attention_heads: int = 16,
                 intermediate_size: int = 4096, num_query_tokens: int = 8, field_embedding_dim: int = 1024,
                 num_fields: int = None, dropout: float = 0.2):
        super().__init__(hidden_size, num_hidden_layers, num_attention_heads, intermediate_size, num_query_tokens,
                         field_embedding_dim, num_fields, dropout)
