from quadruplet_sentence_transformer_amd.evaluation import (InformationRetrievalEvaluator,  # noqa: F401
                                                            SentenceEvaluator, SequentialEvaluator,
                                                            SimilarityFunction, TripletEvaluator)
