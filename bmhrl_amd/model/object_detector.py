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

WIDTH, QUERIES, HEADS, FF_WIDTH, DEPTH = 256, 100, 4, 2048, 6     # fixed in the reference (:13-14, :19-30)


class ObjectDetect(nn.Module):

    def __init__(self, cfg, voc_size):
        super().__init__()
        p_drop = cfg.dout_p
        self.d_model, self.num_classes = cfg.d_model, voc_size + 1          # the extra class is "no object"
        # (modules are registered in the reference's order: the state dict lists its keys the same way)
        self.class_embed = nn.Linear(WIDTH, self.num_classes)
        self.query_embed = nn.Embedding(QUERIES, WIDTH)
        self.pos_enc = PositionalEncoder(WIDTH, p_drop)
        self.input_projection = nn.Linear(cfg.d_model, WIDTH)
        self.encoder = TransformerEncoder(TransformerEncoderLayer(WIDTH, HEADS, FF_WIDTH, p_drop, "relu", normalize_before=True),
                                          DEPTH, nn.LayerNorm(WIDTH), cfg, return_intermediate=False)
        self.linear = nn.Linear(WIDTH, voc_size)                             # in the checkpoint; nobody applies it
        self.decoder = TransformerDecoder(TransformerDecoderLayer(WIDTH, HEADS, WIDTH, cfg.rl_goal_d, FF_WIDTH, p_drop, "relu",
                                                                  normalize_before=True),
                                          DEPTH, nn.LayerNorm(WIDTH), return_intermediate=False)

    def forward(self, feats, key_mask):
        """reference :33-46 -> (class logits (B, 100, V + 1), detached query states (B, 100, 256), "no object" mask (B, 100))"""
        x = LinearFn.apply(feats, self.input_projection.weight, self.input_projection.bias, False, 0.0)
        mem = self.encoder(x, key_mask, self.pos_enc)
        queries = self.query_embed.weight.expand(x.shape[0], QUERIES, WIDTH).contiguous()     # one copy of the table per sample
        states = self.decoder(torch.zeros_like(queries), mem, key_mask, self.pos_enc, queries, None, None, None, None, add_pos=True)
        logits = LinearFn.apply(states, self.class_embed.weight, self.class_embed.bias, False, 0.0)
        no_object = logits.argmax(-1) == self.num_classes - 1                # (arg-max of the soft-max, :43, is that of the logits)
        return logits, states.detach(), no_object.detach()
