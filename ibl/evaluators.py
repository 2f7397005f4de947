"""ibl.evaluators — the reference's evaluation surface (ibl/evaluators.py) on the HIP path."""
from openibl_amd.evaluators import (extract_cnn_feature, extract_features, pairwise_distance,
                                    spatial_nms, evaluate_all, recalls_from_topk, Evaluator)

__all__ = ['extract_cnn_feature', 'extract_features', 'pairwise_distance', 'spatial_nms',
           'evaluate_all', 'recalls_from_topk', 'Evaluator']
