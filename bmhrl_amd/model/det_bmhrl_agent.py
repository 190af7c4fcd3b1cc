"""DetrCaption, the reference's DETR-mode agent (model/det_bmhrl_agent.py:12-208), on the HIP kernels -- SURVEY.md section 8(f)
rank 4: the Conv1d('same') + GroupNorm input projection over the time axis (functional.Conv1dSameFn / GroupNormFn,
csrc/conv_gn.hip), the 100-query object detector (model/object_detector.py), the post-norm video encoder and the causal
caption decoder with its detected-object attention (model/encoder.py, model/decoder.py).  Same constructor, attribute names,
state-dict keys and return tuple.  As in the reference, forward() runs with `use_manager = False` (:177): the manager /
critic / goal-attention modules exist for the checkpoint layout and the teach_* switches but take no part in the
prediction, so cfg.pre_goal_attention must be False (the reference's own forward raises a NameError otherwise)."""
import torch
import torch.nn as nn

from ..functional import Conv1dSameFn, GroupNormFn, LinearFn
from .blocks import PositionalEncoder, VocabularyEmbedder
from .bm_hrl_agent import Manager, SegmentCritic
from .decoder import TransformerDecoder, TransformerDecoderLayer
from .encoder import TransformerEncoder, TransformerEncoderLayer
from .multihead_attention import MultiheadedAttention
from .object_detector import ObjectDetect


FF_WIDTH, N_LAYERS, N_SCALES = 2048, 3, 3        # fixed in the reference (:19, :24, :76)


def _decoder_stack(cfg, width, heads):
    """three post-norm decoder layers of caption width `width` over the video memory + their final LayerNorm"""
    layer = TransformerDecoderLayer(cfg.d_model_video, heads, width, cfg.rl_goal_d, FF_WIDTH, cfg.dout_p, "relu", normalize_before=True)
    return layer, nn.LayerNorm(width)


class DetrCaption(nn.Module):

    def __init__(self, cfg, train_dataset):
        super().__init__()
        dev = getattr(cfg, "device", "cpu")
        d_caps, d_goal, heads, p_drop = cfg.d_model_caps, cfg.rl_goal_d, cfg.rl_att_heads, cfg.dout_p
        # plain attributes of the reference's constructor (:16-25, :31, :73-76), kept by name
        self.name, self.att_layers, self.device = "detr_agent", cfg.rl_att_layers, (torch.device(dev) if isinstance(dev, str) else dev)
        self.dim_feedforward, self.num_layers, self.n_time = FF_WIDTH, N_LAYERS, N_SCALES
        self.dif_work_man_feats, self.normalize_before, self.teaching_worker = False, True, True
        self.voc_size, self.d_model, self.n_head = train_dataset.trg_voc_size, cfg.d_model, heads
        self.pre_goal_attention = cfg.pre_goal_attention
        # --- modules, registered in the reference's order (the state dict lists its keys the same way)
        for attr, width in (("pos_enc", cfg.d_model), ("pos_enc_C", d_caps), ("pos_enc_concat", d_caps + d_goal), ("pos_enc_goal", d_goal)):
            setattr(self, attr, PositionalEncoder(width, p_drop))
        self.emb_C = VocabularyEmbedder(self.voc_size, d_caps)
        self.emb_C.init_word_embeddings(train_dataset.train_vocab.vectors, cfg.unfreeze_word_emb)
        self.encoder = TransformerEncoder(TransformerEncoderLayer(cfg.d_model, heads, FF_WIDTH, p_drop, "relu", normalize_before=True),
                                          N_LAYERS, nn.LayerNorm(cfg.d_model), cfg, return_intermediate=False)
        # without pre-goal attention worker and manager share ONE layer prototype (cloned per stack) and ONE final LayerNorm
        # object (:40-44: a checkpoint then holds the same values under worker_decoder.norm.* and manager_decoder.norm.*);
        # with it the worker's stack is d_goal wider
        manager_proto = _decoder_stack(cfg, d_caps, heads)
        worker_proto = _decoder_stack(cfg, d_caps + d_goal, heads) if self.pre_goal_attention else manager_proto
        self.linear = nn.Linear(worker_proto[1].normalized_shape[0], self.voc_size)
        self.worker_decoder = TransformerDecoder(worker_proto[0], N_LAYERS, worker_proto[1], return_intermediate=False)
        self.manager_decoder = TransformerDecoder(manager_proto[0], N_LAYERS, manager_proto[1], return_intermediate=False)
        self.manager_core = nn.Identity()
        self.manager = Manager(self.device, d_caps, d_goal, p_drop, self.manager_core)
        self.activation = nn.LogSoftmax(dim=-1)
        self.goal_norm, self.goal_dropout = nn.LayerNorm(d_caps), nn.Dropout(p_drop)
        self.goal_attention = MultiheadedAttention(d_caps, d_goal, d_goal, heads, p_drop, cfg.d_model)
        self.goal_feature_attention = MultiheadedAttention(d_goal, d_caps, d_caps, heads, p_drop, cfg.d_model)
        self.manager_modules = [self.manager_core, self.manager, self.manager_decoder]
        self.worker_modules = [self.worker_decoder, self.linear]
        self.query_embed = nn.Embedding(80, 300)                      # (a checkpoint key; the object detector has its own queries)
        self.object_detector = ObjectDetect(cfg, self.voc_size)
        # input projection: kernel sizes 3, 6, 9 over time, each followed by GroupNorm(32)
        self.input_proj = nn.ModuleList(nn.Sequential(nn.Conv1d(cfg.d_model, cfg.d_model, kernel_size=3 * scale, padding="same"),
                                                      nn.GroupNorm(32, cfg.d_model)) for scale in range(1, N_SCALES + 1))
        self._reset_parameters()
        self.critic = SegmentCritic(cfg)                              # (after the Xavier pass, as in the reference: it loads its own weights)
        self.critic_score_threshhold = cfg.rl_critic_score_threshhold
        for block in self.input_proj:
            nn.init.xavier_uniform_(block[0].weight, gain=1)
            nn.init.zeros_(block[0].bias)

    def _checkpoint(self, checkpoint_dir):
        return f"{checkpoint_dir}/{self.name}.pt"

    def save_model(self, checkpoint_dir):
        torch.save(self.state_dict(), self._checkpoint(checkpoint_dir))

    def load_model(self, checkpoint_dir):
        self.load_state_dict(torch.load(self._checkpoint(checkpoint_dir)), strict=False)

    def _phase(self, worker: bool):
        """teach_worker / teach_manager of the reference (:104-117): which side trains, and whether the manager explores"""
        self.warmstarting, self.teaching_worker = False, worker
        for modules, on in ((self.worker_modules, worker), (self.manager_modules, not worker)):
            for module in modules:
                for param in module.parameters():
                    param.requires_grad = on
        self.manager.exploration = not worker

    def teach_worker(self):
        self._phase(True)

    def teach_manager(self):
        self._phase(False)

    def set_inference_mode(self, inference):
        self.manager.exploration = not inference

    def inference(self, x, trg, mask, worker_hid=None, manager_hid=None):
        return self.forward(x, trg, mask)[0], None, None

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def project_input(self, x_video):
        """the three Conv1d('same') + GroupNorm(32) blocks over time (:169-174), activations kept as (B, T, C)"""
        vf = x_video
        for proj in self.input_proj:
            conv, gn = proj[0], proj[1]
            vf = Conv1dSameFn.apply(vf, conv.weight, conv.bias)
            vf = GroupNormFn.apply(vf, gn.weight, gn.bias, gn.num_groups, gn.eps)
        return vf

    def forward(self, x, trg, masks, mode="train"):
        """reference :158-208 -> (log-probs (B, L, V), worker features[..., :300], encoder memory, None, None, class logits)"""
        if self.pre_goal_attention:
            raise NotImplementedError("pre_goal_attention needs the manager branch, which the reference's forward switches off "
                                      "(use_manager = False, model/det_bmhrl_agent.py:177: its own forward raises there)")
        frames = self.project_input(x[0])                             # (the audio half of x is not used in this mode)
        tokens = trg.masked_fill(trg == 3, 1)                         # the end token is embedded as padding (:161-162)
        v_mask = masks["V_mask"]
        class_logits, objects, no_object = self.object_detector(frames, v_mask)
        memory = self.encoder(frames, v_mask, self.pos_enc)
        feats = self.worker_decoder(self.emb_C(tokens), memory, v_mask, self.pos_enc, self.pos_enc_C, masks["C_mask"], None, None, None,
                                    detected_objects=objects, obj_mask=no_object)
        log_probs = torch.log_softmax(LinearFn.apply(feats, self.linear.weight, self.linear.bias, False, 0.0), dim=-1)
        return log_probs, feats[:, :, :300], memory, None, None, class_logits
