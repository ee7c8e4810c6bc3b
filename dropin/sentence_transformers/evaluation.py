from quadruplet_sentence_transformer_amd.evaluation import (SentenceEvaluator, SequentialEvaluator,  # noqa: F401
                                                            SimilarityFunction, TripletEvaluator)
