from quadruplet_sentence_transformer_amd.util import batch_to_device, cos_sim, dot_score, pytorch_cos_sim  # noqa: F401
