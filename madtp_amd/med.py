"""Drop-in for the reference's models/med.py: same class names (encoders, and the teacher-forced answer decoder)."""
from .bert import (BertConfig, BertEmbeddings, BertSelfAttention, BertSelfOutput, BertAttention,  # noqa: F401
                   BertIntermediate, BertOutput)
from .bert import MedBertLayer as BertLayer, MedBertEncoder as BertEncoder, MedBertModel as BertModel  # noqa: F401
from .bert import BertPredictionHeadTransform, BertLMPredictionHead, BertOnlyMLMHead, BertLMHeadModel  # noqa: F401
