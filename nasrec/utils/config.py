from nasrec_amd.utils.config import (MAX_NUM_EMBEDDINGS, NUM_EMBEDDINGS_AVAZU, NUM_EMBEDDINGS_CRITEO, NUM_EMBEDDINGS_KDD,  # noqa: F401
                                     NUM_EMBEDDINGS_TEST)
