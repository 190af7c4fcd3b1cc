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


class DetrCaption(nn.Module):

    def __init__(self, cfg, train_dataset):
        super().__init__()
        self.name = "detr_agent"
        self.att_layers = cfg.rl_att_layers
        self.device = torch.device(cfg.device) if isinstance(getattr(cfg, "device", None), str) else getattr(cfg, "device", "cpu")
        self.dim_feedforward = 2048
        self.dif_work_man_feats = False
        self.voc_size = train_dataset.trg_voc_size
        self.d_model = cfg.d_model
        self.normalize_before = True
        self.num_layers = 3
        self.pre_goal_attention = cfg.pre_goal_attention
        self.pos_enc = PositionalEncoder(cfg.d_model, cfg.dout_p)
        self.pos_enc_C = PositionalEncoder(cfg.d_model_caps, cfg.dout_p)
        self.pos_enc_concat = PositionalEncoder(cfg.d_model_caps + cfg.rl_goal_d, cfg.dout_p)
        self.pos_enc_goal = PositionalEncoder(cfg.rl_goal_d, cfg.dout_p)
        self.n_head = cfg.rl_att_heads
        self.emb_C = VocabularyEmbedder(self.voc_size, cfg.d_model_caps)
        self.emb_C.init_word_embeddings(train_dataset.train_vocab.vectors, cfg.unfreeze_word_emb)
        encoder_layer = TransformerEncoderLayer(cfg.d_model, self.n_head, self.dim_feedforward, cfg.dout_p, "relu",
                                                normalize_before=self.normalize_before)
        self.encoder = TransformerEncoder(encoder_layer, self.num_layers, nn.LayerNorm(cfg.d_model), cfg,
                                          return_intermediate=self.dif_work_man_feats)
        if not self.pre_goal_attention:
            worker_layer = manager_layer = TransformerDecoderLayer(cfg.d_model_video, self.n_head, cfg.d_model_caps, cfg.rl_goal_d,
                                                                   self.dim_feedforward, cfg.dout_p, "relu",
                                                                   normalize_before=self.normalize_before)
            worker_norm = manager_norm = nn.LayerNorm(cfg.d_model_caps)
            self.linear = nn.Linear(cfg.d_model_caps, self.voc_size)
        else:
            worker_layer = TransformerDecoderLayer(cfg.d_model_video, self.n_head, cfg.d_model_caps + cfg.rl_goal_d, cfg.rl_goal_d,
                                                   self.dim_feedforward, cfg.dout_p, "relu", normalize_before=self.normalize_before)
            manager_layer = TransformerDecoderLayer(cfg.d_model_video, self.n_head, cfg.d_model_caps, cfg.rl_goal_d,
                                                    self.dim_feedforward, cfg.dout_p, "relu", normalize_before=self.normalize_before)
            worker_norm = nn.LayerNorm(cfg.d_model_caps + cfg.rl_goal_d)
            manager_norm = nn.LayerNorm(cfg.d_model_caps)
            self.linear = nn.Linear(cfg.d_model_caps + cfg.rl_goal_d, self.voc_size)
        self.worker_decoder = TransformerDecoder(worker_layer, self.num_layers, worker_norm, return_intermediate=False)
        self.manager_decoder = TransformerDecoder(manager_layer, self.num_layers, manager_norm, return_intermediate=False)
        self.manager_core = nn.Identity()
        self.manager = Manager(self.device, cfg.d_model_caps, cfg.rl_goal_d, cfg.dout_p, self.manager_core)
        self.activation = nn.LogSoftmax(dim=-1)
        self.goal_norm = nn.LayerNorm(cfg.d_model_caps)
        self.goal_dropout = nn.Dropout(cfg.dout_p)
        self.goal_attention = MultiheadedAttention(cfg.d_model_caps, cfg.rl_goal_d, cfg.rl_goal_d, self.n_head, cfg.dout_p, cfg.d_model)
        self.goal_feature_attention = MultiheadedAttention(cfg.rl_goal_d, cfg.d_model_caps, cfg.d_model_caps, self.n_head,
                                                           cfg.dout_p, cfg.d_model)
        self.manager_modules = [self.manager_core, self.manager, self.manager_decoder]
        self.worker_modules = [self.worker_decoder, self.linear]
        self.query_embed = nn.Embedding(80, 300)
        self.teaching_worker = True
        self.n_time = 3
        self.object_detector = ObjectDetect(cfg, self.voc_size)
        self.input_proj = nn.ModuleList([nn.Sequential(nn.Conv1d(cfg.d_model, cfg.d_model, kernel_size=3 * i, padding="same"),
                                                       nn.GroupNorm(32, cfg.d_model)) for i in range(1, self.n_time + 1)])
        self._reset_parameters()
        self.critic = SegmentCritic(cfg)
        self.critic_score_threshhold = cfg.rl_critic_score_threshhold
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)

    def save_model(self, checkpoint_dir):
        torch.save(self.state_dict(), checkpoint_dir + f"/{self.name}.pt")

    def load_model(self, checkpoint_dir):
        self.load_state_dict(torch.load(checkpoint_dir + f"/{self.name}.pt"), strict=False)

    def _set_module_grads(self, modules, enable):
        for module in modules:
            for _, param in module.named_parameters():
                param.requires_grad = enable

    def teach_worker(self):
        self.warmstarting = False
        self.teaching_worker = True
        self._set_module_grads(self.worker_modules, True)
        self._set_module_grads(self.manager_modules, False)
        self.manager.exploration = False

    def teach_manager(self):
        self.warmstarting = False
        self.teaching_worker = False
        self._set_module_grads(self.worker_modules, False)
        self._set_module_grads(self.manager_modules, True)
        self.manager.exploration = True

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
        x_video, _ = x
        trg = trg.clone()
        trg[trg == 3] = 1
        C = self.emb_C(trg)
        mask = masks["V_mask"]
        x_video = self.project_input(x_video)
        classified_words, hs_ob_det, ob_mask = self.object_detector(x_video, mask)
        memory = self.encoder(x_video, mask, self.pos_enc)
        worker_feat = self.worker_decoder(C, memory, mask, self.pos_enc, self.pos_enc_C, masks["C_mask"], None, None, None,
                                          detected_objects=hs_ob_det, obj_mask=ob_mask)
        pred = torch.log_softmax(LinearFn.apply(worker_feat, self.linear.weight, self.linear.bias, False, 0.0), dim=-1)
        return pred, worker_feat[:, :, :300], memory, None, None, classified_words
