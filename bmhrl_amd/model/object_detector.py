"""ObjectDetect of the reference's DETR mode (model/object_detector.py:8-46) on the HIP kernels: a 256-wide post-norm
encoder (6 layers) over the projected video features and a decoder (6 layers) of 100 learned queries whose outputs are
classified into the vocabulary; same attribute names and state-dict keys.  (The reference's file imports
torchvision.models.VisionTransformer and never uses it.)"""
import torch
import torch.nn as nn

from ..functional import LinearFn
from .blocks import PositionalEncoder
from .decoder import TransformerDecoder, TransformerDecoderLayer
from .encoder import TransformerEncoder, TransformerEncoderLayer


class ObjectDetect(nn.Module):

    def __init__(self, cfg, voc_size):
        super().__init__()
        self.d_model = cfg.d_model
        hidden_dim = 256
        num_queries = 100
        self.num_classes = voc_size + 1
        self.class_embed = nn.Linear(hidden_dim, self.num_classes)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.pos_enc = PositionalEncoder(hidden_dim, cfg.dout_p)
        self.input_projection = nn.Linear(self.d_model, hidden_dim)
        encoder_layer = TransformerEncoderLayer(hidden_dim, 4, 2048, cfg.dout_p, "relu", normalize_before=True)
        self.encoder = TransformerEncoder(encoder_layer, 6, nn.LayerNorm(hidden_dim), cfg, return_intermediate=False)
        decoder_layer = TransformerDecoderLayer(hidden_dim, 4, hidden_dim, cfg.rl_goal_d, 2048, cfg.dout_p, "relu",
                                                normalize_before=True)
        self.linear = nn.Linear(hidden_dim, voc_size)                  # (never applied by the reference either)
        self.decoder = TransformerDecoder(decoder_layer, 6, nn.LayerNorm(hidden_dim), return_intermediate=False)

    def forward(self, samples, mask):
        """reference :33-46 -> (class logits (B, 100, V + 1), detached query states (B, 100, 256), "no object" mask (B, 100))"""
        samples = LinearFn.apply(samples, self.input_projection.weight, self.input_projection.bias, False, 0.0)
        bs = samples.shape[0]
        memory = self.encoder(samples, mask, self.pos_enc)
        query_pos = self.query_embed.weight.unsqueeze(0).repeat(bs, 1, 1)
        tgt = torch.zeros_like(query_pos)
        hs = self.decoder(tgt, memory, mask, self.pos_enc, query_pos, None, None, None, None, add_pos=True)
        predicted_words = LinearFn.apply(hs, self.class_embed.weight, self.class_embed.bias, False, 0.0)
        # argmax of the softmax == argmax of the logits (:43)
        attention_mask = torch.argmax(predicted_words, -1) == (self.num_classes - 1)
        return predicted_words, hs.detach(), attention_mask.detach()
