"""Drop-in for the reference's models/med.py (encoder side): same class names."""
from .bert import (BertConfig, BertEmbeddings, BertSelfAttention, BertSelfOutput, BertAttention,  # noqa: F401
                   BertIntermediate, BertOutput)
from .bert import MedBertLayer as BertLayer, MedBertEncoder as BertEncoder, MedBertModel as BertModel  # noqa: F401
