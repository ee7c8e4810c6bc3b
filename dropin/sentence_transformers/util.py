from quadruplet_sentence_transformer_amd.util import (batch_to_device, cos_sim, dot_score,  # noqa: F401
                                                      mine_hard_negatives, pytorch_cos_sim, topk_rows, topk_scores)
