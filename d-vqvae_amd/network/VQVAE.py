"""VQVAE wrapper (mirrors DVQ-VAE/network/VQVAE.py:11-53): owns one submodule
`vector_quantization`; forward -> (loss, z_q, perplexity), inference -> (indices, z_q),
get_embbeding(index, dim) -> rows.  The unused constructor arguments are kept."""
import torch.nn as nn

from network.vqvae.quantizer import VectorQuantizer


class VQVAE(nn.Module):
    def __init__(self, h_dim, res_h_dim, n_res_layers, n_embeddings, embedding_dim, beta, a=1,
                 save_img_embedding_map=False):
        super().__init__()
        self.vector_quantization = VectorQuantizer(n_embeddings, embedding_dim, beta, al=a)
        self.img_to_embedding_map = {i: [] for i in range(n_embeddings)} if save_img_embedding_map else None

    def forward(self, inputs, verbose=False):
        assert not verbose
        embedding_loss, z_q, perplexity, _, _ = self.vector_quantization(inputs, True)
        return embedding_loss, z_q, perplexity

    def inference(self, inputs, verbose=False):
        return self.vector_quantization(inputs, False)

    def get_embbeding(self, index, dim):
        return self.vector_quantization.get_emb(index, dim)
