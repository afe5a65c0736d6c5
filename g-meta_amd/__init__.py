"""gmeta_amd -- MI355X-native (gfx950) implementation of G-Meta's inner-loop hot path.

Host mirror of the reference's Python surface (same names, argument meaning and error behaviour)
over the C-ABI library libgmeta_hip.so (include/gmeta_hip.h):

    reference (G-Meta/)                          here
    subgraph_data_processing.Subgraphs/collate   gmeta_amd.subgraphs.Subgraphs / collate
    learner.Classifier                           gmeta_amd.learner.Classifier
    meta.Meta (.forward / .finetunning)          gmeta_amd.meta.Meta

There is no CPU fallback: every compute entry point raises if the HIP library or a GPU is missing.
"""
from . import _lib  # noqa: F401
from .graphstore import GraphStore  # noqa: F401
from .subgraphs import Subgraphs, SubgraphBatch, collate  # noqa: F401
from .learner import Classifier  # noqa: F401
from .meta import Meta  # noqa: F401

__all__ = ['GraphStore', 'Subgraphs', 'SubgraphBatch', 'collate', 'Classifier', 'Meta']
