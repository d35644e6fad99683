"""Beam-search decoding — mirror of fairseq/sequence_generator.py (SequenceGenerator._generate :179-541,
finalize_hypos :575-696, EnsembleModel.forward_encoder/forward_decoder :800-868) and fairseq/search.py BeamSearch.step
(:109-144), for one model.  Decoder steps run through the incremental-state path of the HIP modules (K/V caches kept
batch-major [B*beam, T, C]; single-query fused attention).

Differences that do not change results: finished sentences are masked out instead of being removed from the batch
(the reference shrinks the batch, :427-463 — an optimisation only; every sentence's search is independent)."""
import math
from typing import Dict, List, Optional

import torch
from torch import Tensor


class BeamSearch:
    """search.py:100-144."""

    def __init__(self, tgt_dict):
        self.pad, self.unk, self.eos = tgt_dict.pad(), tgt_dict.unk(), tgt_dict.eos()
        self.vocab_size = len(tgt_dict)

    def step(self, step: int, lprobs, scores):
        bsz, beam_size, vocab_size = lprobs.size()
        if step == 0:
            lprobs = lprobs[:, ::beam_size, :].contiguous()  # all hypotheses equal at step 0: use the first beam only
        else:
            lprobs = lprobs + scores[:, :, step - 1].unsqueeze(-1)
        top = torch.topk(lprobs.view(bsz, -1), k=min(beam_size * 2, lprobs.view(bsz, -1).size(1) - 1))
        scores_buf, indices_buf = top[0], top[1]
        beams_buf = indices_buf // vocab_size
        indices_buf = indices_buf.fmod(vocab_size)
        return scores_buf, indices_buf, beams_buf


class SequenceGenerator:
    def __init__(self, models, tgt_dict, beam_size=1, max_len_a=0, max_len_b=200, min_len=1, normalize_scores=True,
                 len_penalty=1.0, unk_penalty=0.0, temperature=1.0, match_source_len=False, no_repeat_ngram_size=0,
                 search_strategy=None, eos=None, fused=True, use_graph=True, cross_kernel=None):
        self.model = models[0] if isinstance(models, (list, tuple)) else models
        self.tgt_dict = tgt_dict
        self.pad, self.unk = tgt_dict.pad(), tgt_dict.unk()
        self.eos = tgt_dict.eos() if eos is None else eos
        self.vocab_size = len(tgt_dict)
        self.beam_size = min(beam_size, self.vocab_size - 1)
        self.max_len_a, self.max_len_b, self.min_len = max_len_a, max_len_b, min_len
        self.normalize_scores, self.len_penalty, self.unk_penalty = normalize_scores, len_penalty, unk_penalty
        self.temperature = temperature
        assert temperature > 0 and not match_source_len and no_repeat_ngram_size == 0
        self.search = BeamSearch(tgt_dict) if search_strategy is None else search_strategy
        # fused=True (default): the device-resident loop of decode_engine.py (one captured HIP graph per step, no per-step host
        # sync); fused=False: the module-by-module mirror of the reference loop below (same kernels, host-driven) — kept as the
        # readable restatement and as the cross-check of the engine.  A custom search strategy needs the host loop.
        self.fused = bool(fused) and search_strategy is None and (eos is None or eos == tgt_dict.eos())
        self._engine = None
        self.use_graph, self.cross_kernel = use_graph, cross_kernel
        self.model.eval()

    @torch.no_grad()
    def generate(self, models, sample, prefix_tokens=None, **kwargs):
        assert prefix_tokens is None
        return self._generate(sample)

    def _forward_decoder(self, tokens, encoder_out, incremental_state):
        """sequence_generator.py:806-868 for a single model: last-step logits -> fp32 log-softmax (temperature applied)."""
        logits, _ = self.model.decoder.forward(tokens, encoder_out=encoder_out, incremental_state=incremental_state)
        logits = logits[:, -1:, :]
        if self.temperature != 1.0:
            logits = logits / self.temperature
        return self.model.get_normalized_probs((logits, None), log_probs=True)[:, -1, :]

    def _generate(self, sample):
        net_input = sample["net_input"]
        src_tokens = net_input["src_tokens"]
        bsz, src_len = src_tokens.size()[:2]
        beam_size = self.beam_size
        device = src_tokens.device
        max_len = min(int(self.max_len_a * src_len + self.max_len_b), self.model.max_decoder_positions() - 1)
        assert self.min_len <= max_len
        encoder_out = self.model.encoder.forward_torchscript(net_input)
        if self.fused:
            from .decode_engine import BeamDecodeEngine
            if BeamDecodeEngine.supported(self.model.decoder):
                if self._engine is None or self._engine.max_len != max_len:
                    self._engine = BeamDecodeEngine(self.model.decoder, self.tgt_dict, beam_size, max_len, self.min_len,
                                                    self.normalize_scores, self.len_penalty, self.unk_penalty, self.temperature,
                                                    use_graph=self.use_graph, cross_kernel=self.cross_kernel)
                return self._engine.generate(encoder_out, bsz)
        new_order = torch.arange(bsz, device=device).view(-1, 1).repeat(1, beam_size).view(-1)
        encoder_out = self.model.encoder.reorder_encoder_out(encoder_out, new_order)
        incremental_state: Dict[str, Dict[str, Optional[Tensor]]] = {}

        scores = torch.zeros(bsz * beam_size, max_len + 1, device=device, dtype=torch.float32)
        tokens = torch.full((bsz * beam_size, max_len + 2), self.pad, device=device, dtype=torch.long)
        tokens[:, 0] = self.eos
        cands_to_ignore = torch.zeros(bsz, beam_size, device=device).eq(-1)
        finalized: List[List[Dict[str, Tensor]]] = [[] for _ in range(bsz)]
        finished = [False] * bsz
        num_remaining = bsz
        cand_size = 2 * beam_size
        bbsz_offsets = (torch.arange(0, bsz, device=device) * beam_size).unsqueeze(1)
        cand_offsets = torch.arange(0, cand_size, device=device)
        reorder_state = None

        for step in range(max_len + 1):
            if reorder_state is not None:
                self.model.decoder.reorder_incremental_state_scripting(incremental_state, reorder_state)
                encoder_out = self.model.encoder.reorder_encoder_out(encoder_out, reorder_state)
            lprobs = self._forward_decoder(tokens[:, :step + 1], encoder_out, incremental_state)
            lprobs[lprobs != lprobs] = -math.inf
            lprobs[:, self.pad] = -math.inf
            lprobs[:, self.unk] -= self.unk_penalty
            if step >= max_len:
                lprobs[:, :self.eos] = -math.inf
                lprobs[:, self.eos + 1:] = -math.inf
            if step < self.min_len:
                lprobs[:, self.eos] = -math.inf
            cand_scores, cand_indices, cand_beams = self.search.step(
                step, lprobs.view(bsz, -1, self.vocab_size), scores.view(bsz, beam_size, -1)[:, :, :step])
            cand_bbsz_idx = cand_beams.add(bbsz_offsets)
            eos_mask = cand_indices.eq(self.eos) & cand_scores.ne(-math.inf)
            eos_mask[:, :beam_size][cands_to_ignore] = False
            # finalize hypotheses whose eos is among the top beam_size candidates (:385-416)
            top_eos = eos_mask[:, :beam_size]
            if top_eos.any():
                for sent, col in top_eos.nonzero(as_tuple=False).tolist():
                    if finished[sent] or len(finalized[sent]) >= beam_size:
                        continue
                    bi = int(cand_bbsz_idx[sent, col])
                    sc = cand_scores[sent, col].clone()
                    toks = tokens[bi, 1:step + 2].clone()
                    toks[step] = self.eos
                    pos = scores[bi, :step + 1].clone()
                    pos[step] = sc
                    pos[1:] = pos[1:] - pos[:-1].clone()
                    if self.normalize_scores:
                        sc = sc / (step + 1) ** self.len_penalty
                    finalized[sent].append({"tokens": toks, "score": sc, "attention": None, "alignment": None,
                                            "positional_scores": pos})
                for sent in set(s for s, _ in top_eos.nonzero(as_tuple=False).tolist()):
                    if not finished[sent] and (len(finalized[sent]) == beam_size or step == max_len):
                        finished[sent] = True
                        num_remaining -= 1
            if num_remaining == 0 or step >= max_len:
                break
            # choose the first beam_size non-eos candidates as the next active hypotheses (:465-499)
            eos_mask[:, :beam_size] = ~((~cands_to_ignore) & (~eos_mask[:, :beam_size]))
            active_mask = eos_mask.type_as(cand_offsets) * cand_size + cand_offsets[:eos_mask.size(1)]
            new_cands_to_ignore, active_hypos = torch.topk(active_mask, k=beam_size, dim=1, largest=False)
            cands_to_ignore = new_cands_to_ignore.ge(cand_size)[:, :beam_size]
            active_bbsz_idx = torch.gather(cand_bbsz_idx, dim=1, index=active_hypos).view(-1)
            tokens[:, :step + 1] = torch.index_select(tokens[:, :step + 1], dim=0, index=active_bbsz_idx)
            tokens.view(bsz, beam_size, -1)[:, :, step + 1] = torch.gather(cand_indices, dim=1, index=active_hypos)
            if step > 0:
                scores[:, :step] = torch.index_select(scores[:, :step], dim=0, index=active_bbsz_idx)
            scores.view(bsz, beam_size, -1)[:, :, step] = torch.gather(cand_scores, dim=1, index=active_hypos)
            reorder_state = active_bbsz_idx

        for sent in range(bsz):
            sc = torch.tensor([float(h["score"]) for h in finalized[sent]])
            _, order = torch.sort(sc, descending=True)
            finalized[sent] = [finalized[sent][i] for i in order.tolist()]
        return finalized
