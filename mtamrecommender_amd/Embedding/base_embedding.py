"""Base class of the embedding module (reference: Embedding/base_embedding.py:7-60)."""
import numpy as np

from ..util.model_log import create_log


class Base_embedding(object):

    def __init__(self, is_training=True, config_file=None):
        self.embedding_file_path = config_file
        self.is_training = is_training
        self.logger = create_log().logger
        self.init_placeholders()

    def init_placeholders(self):
        pass

    def get_embedding(self):
        pass

    def make_feed_dic(self, batch_data):
        pass

    def init_embedding_lookup_table(self, name, total_count, embedding_dim, is_training=True, rng=None):
        """[total_count, embedding_dim] table, U(-r, r) with r = sqrt(6 / embedding_dim)
        (Embedding/base_embedding.py:46-60)."""
        total_count, embedding_dim = int(total_count), int(embedding_dim)
        r = float(np.sqrt(np.float32(6.0 / embedding_dim)))
        rng = rng if rng is not None else np.random.Generator(np.random.PCG64(1234))
        return rng.uniform(-r, r, size=(total_count, embedding_dim)).astype(np.float32)
