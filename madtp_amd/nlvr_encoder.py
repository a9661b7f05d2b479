"""Drop-in for the reference's models/nlvr_encoder.py: same class names."""
from .bert import (BertConfig, BertEmbeddings, BertSelfAttention, BertSelfOutput, BertAttention,  # noqa: F401
                   BertIntermediate, BertOutput)
from .bert import NlvrBertLayer as BertLayer, NlvrBertEncoder as BertEncoder, NlvrBertModel as BertModel  # noqa: F401
