"""Token -> waveform on the GPU.

Mirror of the reference's ``S3Token2Wav.inference`` (src/chatterbox_vllm/models/s3gen/s3gen.py:297-317) for a pre-computed
reference dict (``ref_dict``, the "prod API" of s3gen.py:176-186, as ``ChatterboxTTS`` passes it, tts.py:489-493):

    speech tokens --FlowEncoder--> mu, spks, cond --CausalConditionalCFM--> mel --HiFTGenerator--> waveform --trim fade-->

``decoder`` may replace the flow-matching stage by any callable with the reference decoder's call signature
(flow.py:279-285): ``decoder(mu=[1, 80, L], mask=[1, 1, L], spks=[1, 80], cond=[1, 80, L], n_timesteps=int) ->
(mel [1, 80, L], _)``, e.g. the reference's own ``s3gen.flow.decoder`` where its libraries are installed.  The default is this
package's ``CausalConditionalCFM`` (cfm_decoder.py), built from the ``decoder.estimator.*`` entries of the flow state dict.

``inference_batch`` takes several utterances of one voice through the flow-matching stage in shared kernel launches (the
reference runs them one after the other, tts.py:489-493 inside its loop over the batch): the stage's launches are small for
one short utterance, and several utterances together fill the GPU.

``chains`` > 1 builds that many copies of the whole chain (weights, workspaces), each driven from its own host thread on
its own stream by ``render_many``: while one copy's flow-matching stage runs, the short launches of another copy's encoder
and vocoder fill the CUs it leaves idle.  Measured on the reference's 154-chunk benchmark job: one utterance at a time
17.3 s, eight per call on one chain 13.7 s, two chains of four per call 12.9 s (the default the facade builds)."""
from __future__ import annotations

import queue
import threading
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from .cfm_decoder import CausalConditionalCFM
from .flow_encoder import FlowEncoder
from .hift import HiFTGenerator

S3GEN_SR = 24000


class S3Token2Wav:
    def __init__(self, flow_state_dict: Dict[str, torch.Tensor], mel2wav_state_dict: Dict[str, torch.Tensor],
                 decoder: Optional[Callable] = None, max_tokens: int = 1500, device: int = 0, max_batch: int = 1,
                 chains: int = 1, split_bf16=False):
        self.device = torch.device("cuda", device)
        self.encoder = FlowEncoder(flow_state_dict, max_tokens=max_tokens, device=device)
        self.mel2wav = HiFTGenerator(mel2wav_state_dict, max_batch=1, max_frames=2 * max_tokens, device=device)
        self._own_decoder = decoder is None
        # split_bf16: 1 / True = the decoder's attention launches on bf16 triples, 2 = its linear layers and convolutions too
        # (CausalConditionalCFM; 0 / False = the fp32 matrix instruction)
        self.decoder = (CausalConditionalCFM(flow_state_dict, max_frames=2 * max_tokens, device=device, max_batch=max_batch,
                                             split_bf16=split_bf16)
                        if decoder is None else decoder)
        self.max_batch = max_batch
        # silence the first 20 ms and fade in over the next 20 ms (s3gen.py:247-251)
        n_trim = S3GEN_SR // 50
        fade = torch.zeros(2 * n_trim)
        fade[n_trim:] = (torch.cos(torch.linspace(torch.pi, 0, n_trim)) + 1) / 2
        self.trim_fade = fade.to(self.device)
        # further copies of the chain for render_many; they slice the same fixed noise, so every copy returns the same bits
        self.siblings: List["S3Token2Wav"] = []
        if chains > 1 and decoder is None:
            for _ in range(chains - 1):
                sib = S3Token2Wav(flow_state_dict, mel2wav_state_dict, None, max_tokens, device, max_batch, 1, split_bf16)
                sib.decoder.rand_noise = self.decoder.rand_noise
                self.siblings.append(sib)
        self._streams = [torch.cuda.Stream(device=self.device) for _ in self.siblings]

    @torch.inference_mode()
    def flow_inference(self, speech_tokens: torch.Tensor, ref_dict: dict, n_timesteps: int = 10) -> torch.Tensor:
        """s3gen.py:277-288 with finalize=True: mel [1, 80, 2 T_new]."""
        if speech_tokens.dim() == 1:
            speech_tokens = speech_tokens.unsqueeze(0)
        d = self.encoder.decoder_inputs(speech_tokens, ref_dict["prompt_token"], ref_dict["prompt_feat"], ref_dict["embedding"])
        mel_len1 = d.pop("prompt_len")
        feat, _ = self.decoder(mu=d["mu"], mask=d["mask"], spks=d["spks"], cond=d["cond"], n_timesteps=n_timesteps)
        feat = feat[:, :, mel_len1:]
        assert feat.shape[2] == 2 * speech_tokens.shape[1], "the decoder must return one mel frame per mu column"
        return feat.float()

    @torch.inference_mode()
    def hift_inference(self, speech_feat: torch.Tensor, cache_source: Optional[torch.Tensor] = None, seed: int = 0):
        """s3gen.py:290-294."""
        return self.mel2wav.inference(speech_feat, cache_source, seed=seed)

    @torch.inference_mode()
    def inference(self, speech_tokens: torch.Tensor, ref_dict: dict, cache_source: Optional[torch.Tensor] = None,
                  no_trim: bool = False, n_timesteps: int = 10, seed: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        """s3gen.py:296-317: (waveform [1, 480 * 2 T_new], source)."""
        mels = self.flow_inference(speech_tokens, ref_dict, n_timesteps)
        wav, src = self.hift_inference(mels, cache_source, seed=seed)
        if not no_trim:
            n = min(wav.shape[1], self.trim_fade.numel())
            wav[:, :n] *= self.trim_fade[:n]
        return wav, src

    @torch.inference_mode()
    def inference_batch(self, speech_tokens: Sequence[torch.Tensor], ref_dict: dict, no_trim: bool = False, n_timesteps: int = 10,
                        seed: int = 0) -> List[torch.Tensor]:
        """``inference`` for several utterances of one voice (at most ``max_batch``): list of waveforms [1, 480 * 2 T_u].
        Needs this package's decoder (``forward_batch``)."""
        if not hasattr(self.decoder, "forward_batch"):
            raise TypeError("the decoder in use has no forward_batch: call inference per utterance")
        ins = []
        for tok in speech_tokens:
            tok = tok.unsqueeze(0) if tok.dim() == 1 else tok
            ins.append(self.encoder.decoder_inputs(tok, ref_dict["prompt_token"], ref_dict["prompt_feat"], ref_dict["embedding"]))
        mels = self.decoder.forward_batch([d["mu"] for d in ins], [d["spks"] for d in ins], [d["cond"] for d in ins],
                                          n_timesteps=n_timesteps)
        wavs = []
        for d, mel in zip(ins, mels):
            wav, _ = self.hift_inference(mel[:, :, d["prompt_len"]:].float().contiguous(), None, seed=seed)
            if not no_trim:
                n = min(wav.shape[1], self.trim_fade.numel())
                wav[:, :n] *= self.trim_fade[:n]
            wavs.append(wav)
        return wavs

    @torch.inference_mode()
    def render_many(self, speech_tokens: Sequence[torch.Tensor], ref_dict: dict, n_timesteps: int = 10, seed: int = 0,
                    no_trim: bool = False) -> List[torch.Tensor]:
        """Waveforms [1, 480 * 2 T_u] of any number of utterances of one voice, in input order: longest first, ``max_batch``
        per call, the calls handed to the chain's copies as they become free (``chains`` of them, one host thread and one
        stream each).  With one copy and ``max_batch`` 1 this is the reference's loop (tts.py:489-493)."""
        out: List[Optional[torch.Tensor]] = [None] * len(speech_tokens)
        order = sorted((i for i, t in enumerate(speech_tokens) if t.numel() > 0), key=lambda i: -speech_tokens[i].numel())
        per_call = self.max_batch if hasattr(self.decoder, "forward_batch") else 1
        groups = [order[j:j + per_call] for j in range(0, len(order), per_call)]

        def run(chain: "S3Token2Wav", group: List[int]) -> None:
            toks = [speech_tokens[i].to(chain.device) for i in group]
            if per_call == 1:
                wavs = [chain.inference(toks[0], ref_dict, no_trim=no_trim, n_timesteps=n_timesteps, seed=seed)[0]]
            else:
                wavs = chain.inference_batch(toks, ref_dict, no_trim=no_trim, n_timesteps=n_timesteps, seed=seed)
            for i, wav in zip(group, wavs):
                out[i] = wav

        if not self.siblings or len(groups) < 2:
            for g in groups:
                run(self, g)
            return out
        todo: "queue.Queue[List[int]]" = queue.Queue()
        for g in groups:
            todo.put(g)
        errors: List[BaseException] = []
        caller = torch.cuda.current_stream(self.device)

        def work(chain: "S3Token2Wav", stream: Optional[torch.cuda.Stream]) -> None:
            try:
                with torch.inference_mode(), torch.cuda.stream(stream if stream is not None else caller):
                    if stream is not None:
                        stream.wait_stream(caller)            # inputs prepared on the caller's stream
                    while True:
                        try:
                            g = todo.get_nowait()
                        except queue.Empty:
                            break
                        run(chain, g)
                    if stream is not None:
                        stream.synchronize()
            except BaseException as e:                        # surfaced on the calling thread
                errors.append(e)

        threads = [threading.Thread(target=work, args=(c, st)) for c, st in zip(self.siblings, self._streams)]
        for t in threads:
            t.start()
        work(self, None)
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return out

    def shutdown(self) -> None:
        for sib in self.siblings:
            sib.shutdown()
        self.siblings = []
        self.encoder.shutdown()
        self.mel2wav.shutdown()
        if self._own_decoder:
            self.decoder.shutdown()
