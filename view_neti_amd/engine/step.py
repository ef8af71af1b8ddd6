"""One textual-inversion optimisation step on one GPU as a single replayable launch schedule.

Mirrors the loop body of `Coach.train` (training/coach.py:154-231):

    latents = vae.encode(px).latent_dist.sample() * scaling_factor        -> VAEEncoderEngine + sample_add_noise
    noise, timesteps, noisy = randn_like, randint, scheduler.add_noise    -> device RNG + sample_add_noise
    _hs = get_text_conditioning(...)                                      -> TextEngine.forward
    model_pred = unet(noisy, timesteps, _hs).sample                       -> UNetEngine.forward
    loss = mse_loss(model_pred.float(), target.float())                   -> mse_loss_grad
    accelerator.backward(loss)  (GradScaler-scaled under fp16)            -> UNetEngine.backward + TextEngine.backward
    optimizer.step(); zero_grad()                                         -> adamw_flat (flat bucket)

Everything that varies per step lives in device memory (RNG counter, optimizer step, loss scale,
learning rate), so the step is captured once into a hipGraph and replayed.  Deliberate
deviations from the reference, all documented in DESIGN.md: the placeholder-embedding "restore"
(coach.py:222-229) is dropped (the table is never in the optimizer); noise/timesteps come from a
counter-hash device RNG instead of torch's Philox stream; data-parallel training all-reduces one
flat mapper-gradient bucket instead of wrapping the text encoder in DDP.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import ops
from .. import sd_config as sc
from .text import MapperState, TextEngine
from .unet import UNetEngine
from .vae import VAEEncoderEngine


def alphas_cumprod(cfg: sc.DDPMConfig) -> torch.Tensor:
    """DDPMScheduler(beta_schedule='scaled_linear'): betas = linspace(sqrt(b0), sqrt(b1), T)^2."""
    betas = torch.linspace(cfg.beta_start ** 0.5, cfg.beta_end ** 0.5, cfg.num_train_timesteps,
                           dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class TrainStepEngine:
    def __init__(self, cfg: sc.SDConfig, unet_w: Dict, vae_w: Dict, clip_w: Dict, batch: int, height: int,
                 width: int, mapper_object: Dict[str, torch.Tensor], w_enc_object: torch.Tensor,
                 norm_scale_object: Optional[float], alpha_object: float = 0.2,
                 mapper_view: Optional[Dict[str, torch.Tensor]] = None, w_enc_view: Optional[torch.Tensor] = None,
                 norm_scale_view: Optional[float] = None, alpha_view: float = 0.2, train_view: bool = True,
                 n_view_params: int = 12, lr: float = 1e-3, betas=(0.9, 0.999), adam_eps: float = 1e-8,
                 weight_decay: float = 1e-2, loss_scale: Optional[float] = None, growth_interval: int = 2000,
                 seed: int = 0, world_size: int = 1, device_rng: bool = True, device: str = "cuda",
                 need_backward: bool = True, grad_accum: int = 1, overlap: bool = False,
                 unconstrained_object: bool = False, unconstrained_view: bool = False,
                 nested_dropout_prob: float = 0.0, hidden_object: int = 64,
                 legacy_pe_object: Optional[torch.Tensor] = None, enc_dim_object: int = 64,
                 output_bypass_object: bool = True, output_bypass_view: bool = True, exchange=None,
                 moment_cache_images: int = 0):
        """mapper_object: one mapper state_dict, or a list of them (learnable_mode 3: one object mapper per
        scene, `mapper_object_lookup`, training/coach.py:505-552) — `set_batch(object_index=k)` picks the one
        the batch trains."""
        from .text import flatten_mapper_state
        self.cfg = cfg
        self.B, self.H, self.W = batch, height, width
        self.dev = device
        self.world_size = world_size
        # the exchange step.  On the nccl (= RCCL) backend it is a call into the library's own communicator
        # (vneti_allreduce_flat, csrc/comm.hip): stream-ordered and capturable, so the data-parallel step is the SAME single
        # hipGraph as the one-GPU step plus one collective node (capture()).  VNETI_RCCL_DIRECT=0, the gloo backend (CPU
        # tests, two ranks on one GPU) and any failure to build the communicator fall back to torch.distributed.all_reduce
        # issued from the host between two graphs.  `exchange` may also be handed in (tests: a world-size-1 communicator).
        self.exchange = exchange
        if world_size > 1 and exchange is None:
            import os
            import torch.distributed as dist
            if os.environ.get("VNETI_RCCL_DIRECT", "1") != "0" and dist.is_initialized() and dist.get_backend() == "nccl":
                from ..parallel import enable_direct_rccl
                try:  # the outcome is agreed over the process group (RcclComm): every rank raises, or none does
                    self.exchange = enable_direct_rccl()
                except RuntimeError as err:  # loud, not fatal: torch's communicator does the same collective
                    if os.environ.get("VNETI_REQUIRE_ONE_GRAPH", "0") == "1":
                        raise
                    import warnings
                    warnings.warn(f"library RCCL communicator unavailable ({err}); using torch.distributed.all_reduce")
        self.device_rng = device_rng
        if loss_scale is None:
            # accelerate creates a GradScaler for mixed_precision fp16 only; bf16 has f32's exponent range: a STATIC scale of 1
            # (growth_interval 0: never halved, never grown — the non-finite check and the skip-step logic stay, they
            # cost one tiny kernel)
            from .. import lib
            loss_scale = 1.0 if lib.precision() == "bf16" else 65536.0
            if lib.precision() == "bf16":
                growth_interval = 0
        self.growth_interval = growth_interval
        nlev = len(cfg.vae.block_out_channels)
        self.h, self.w = height >> (nlev - 1), width >> (nlev - 1)
        Lc = cfg.vae.latent_channels
        D = cfg.clip.hidden_size
        # ---- trainable state: one flat f32 bucket [object mapper 0 .. K-1 | view mapper] ----
        objs = list(mapper_object) if isinstance(mapper_object, (list, tuple)) else [mapper_object]
        flats = [flatten_mapper_state(sd) for sd in objs]
        self.n_objects = len(flats)
        self.n_obj = flats[0].numel()  # floats per object mapper
        assert all(f.numel() == self.n_obj for f in flats), "object mappers must share one architecture"
        flat_v = flatten_mapper_state(mapper_view) if (mapper_view is not None and train_view) else None
        n_all_obj = self.n_obj * self.n_objects
        n = n_all_obj + (flat_v.numel() if flat_v is not None else 0)
        self.params = torch.zeros(n, dtype=torch.float32, device=device)
        self.params[:n_all_obj].copy_(torch.cat(flats))
        if flat_v is not None:
            self.params[n_all_obj:].copy_(flat_v)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.n_all_obj = n_all_obj
        # device-side choice of the object mapper (graph-replay safe) + torch's per-parameter Adam step counts
        self.obj_slot = torch.zeros(1, dtype=torch.int32, device=device)
        self.seg_step = torch.zeros(self.n_objects, dtype=torch.int32, device=device)
        self.active_object = 0
        multi = self.n_objects > 1
        # RNG state first: nested dropout draws from it
        self.rng_state = torch.tensor([seed & 0x7FFFFFFF, 0], dtype=torch.int32, device=device)
        # legacy_pe_object: the [1024][2] frequencies of a legacy (arch_view_net <= 14) object mapper; its state dict then
        # carries input_layer.* and enc_dim_object = anchors * layers = 160 (models/neti_mapper.py:90-163)
        mo = MapperState(self.params[:n_all_obj],
                         w_enc_object.to(device).float().contiguous() if legacy_pe_object is None else None,
                         norm_scale_object, alpha_object, hidden=hidden_object, enc_dim=enc_dim_object,
                         unconstrained=unconstrained_object,
                         nested_dropout_prob=nested_dropout_prob, slot=self.obj_slot if multi else None,
                         slot_stride=self.n_obj if multi else 0,
                         legacy_w_pe=(legacy_pe_object.to(device).float().contiguous()
                                      if legacy_pe_object is not None else None),
                         output_bypass=output_bypass_object)
        mv, gv = None, None
        if mapper_view is not None:
            if flat_v is not None:
                pv, gv = self.params[n_all_obj:], self.grads[n_all_obj:]
            else:  # frozen pretrained view mapper (learnable_mode 4/5)
                pv = flatten_mapper_state(mapper_view).to(device)
            mv = MapperState(pv, w_enc_view.to(device).float().contiguous(), norm_scale_view, alpha_view,
                             unconstrained=unconstrained_view,
                             nested_dropout_prob=nested_dropout_prob if flat_v is not None else 0.0,
                             output_bypass=output_bypass_view)
        # ---- device-resident scalars ----
        self.grad_accum = grad_accum
        # accelerate scales each micro-loss by 1/accum and DDP averages over ranks: fold both into AdamW
        self.hyper = torch.tensor([lr, betas[0], betas[1], adam_eps, weight_decay, float(world_size * grad_accum)],
                                  dtype=torch.float32, device=device)
        self.scaler = torch.tensor([loss_scale, 0.0, 0.0], dtype=torch.float32, device=device)
        self.opt_step = torch.zeros(1, dtype=torch.int32, device=device)
        self.loss_sum = torch.zeros(1, dtype=torch.float32, device=device)
        self.ac = alphas_cumprod(cfg.ddpm).to(device)
        # ---- engines ----  (every rank builds these three in this order: the one place where the autotuner's picks are
        # shared by a broadcast — rank-0-only engines such as the validator's stay outside, parallel.shared_picks)
        from ..parallel import shared_picks
        with shared_picks():
            self.unet = UNetEngine(cfg.unet, unet_w, batch, self.h, self.w, cfg.clip.max_positions, device, need_backward)
            self.vae = VAEEncoderEngine(cfg.vae, vae_w, batch, height, width, device)
            self.text = TextEngine(cfg.clip, clip_w, cfg.unet.n_cross_layers, batch, self.unet.timesteps, self.unet.ctx_k,
                                   self.unet.ctx_v, self.unet.dctx_k, self.unet.dctx_v, mo, self.grads[:n_all_obj], mv,
                                   gv, n_view_params, train_view, device, need_backward, rng_state=self.rng_state)
        self.timesteps = self.unet.timesteps
        self.pixel_values = self.vae.x_in
        shape = (batch, Lc, self.h, self.w)
        self.eps = torch.zeros(shape, dtype=torch.float32, device=device)
        self.noise = torch.zeros(shape, dtype=torch.float32, device=device)
        self.latents = torch.zeros(shape, dtype=torch.float32, device=device)
        self.target = torch.zeros(shape, dtype=torch.float32, device=device)
        # ---- optional cache of the VAE posterior moments per dataset image (deterministic datasets only: the CALLER vouches
        # for that — compat/coach.py enables it for augmentation_key 0).  `latent_dist.sample()` is still drawn every step.
        self.n_cache = int(moment_cache_images)
        if self.n_cache:
            hw2 = self.h * self.w
            self.mcache = torch.zeros(self.n_cache, hw2, 2 * Lc, dtype=self.vae.moments.dtype, device=device)
            self.img_idx = torch.zeros(batch, dtype=torch.int64, device=device)
            self._cached_images = set()   # host mirror: which slots hold moments
            self._batch_cached = False    # does the batch set by set_batch() consist of cached images only
            self._batch_images = ()
        self.graph_a_c = self.graph_acc_c = None  # the captured step without the VAE encoder (moments from the cache)
        from .staging import HostStager
        self.stager = HostStager()
        self.need_backward = need_backward
        self.overlap = overlap
        self.side = torch.cuda.Stream() if overlap else None
        self.graph_a = self.graph_b = self.graph_acc = None
        self.exchange_in_graph = False
        self.micro = 0
        self.n_loss = batch * Lc * self.h * self.w

    # ------------------------------------------------------------------ inputs
    def set_batch(self, pixel_values, input_ids, placeholder_object, placeholder_view=None, view_params=None,
                  object_index: int = 0, image_idx=None):
        """object_index: which object mapper this batch trains (a batch is single-scene:
        models/net_clip_text_embedding.py:67-76 asserts one placeholder id and looks its mapper up)."""
        if not 0 <= object_index < self.n_objects:
            raise ValueError(f"object_index {object_index} out of range (have {self.n_objects} object mappers)")
        if self.micro != 0 and object_index != self.active_object:
            raise ValueError("the object mapper may not change inside a gradient-accumulation group")
        self.active_object = object_index
        self.obj_slot.fill_(object_index)
        # every host -> device upload of the batch goes through a pinned staging slot (engine/staging.py): the host does not
        # wait for the previous step's graph, it enqueues the copies behind it and moves on
        st = self.stager
        st.begin()
        try:
            if self.n_cache:
                if image_idx is None:
                    raise ValueError("the moment cache needs the dataset index of every image of the batch (image_idx)")
                ids = tuple(int(i) for i in image_idx)
                if min(ids) < 0 or max(ids) >= self.n_cache:
                    raise ValueError(f"image_idx {ids} outside the moment cache (0..{self.n_cache - 1})")
                st.upload("img_idx", self.img_idx, torch.as_tensor(ids, dtype=torch.int64))
                self._batch_images = ids
                self._batch_cached = all(i in self._cached_images for i in ids)
            self.text.set_batch(input_ids, placeholder_object, placeholder_view, view_params, upload=st.upload)
            if pixel_values is not None:  # None: the device input pipeline already wrote self.pixel_values
                st.upload("pixels", self.pixel_values, pixel_values)  # (large: a blocking copy, issued LAST)
        finally:
            st.end()

    def train(self, mode: bool = True):
        """nested dropout is a training-time feature (neti_mapper.py:403); eval() turns the draws off.
        (a captured graph keeps the mode it was captured in)"""
        self.text.training = mode
        return self

    def object_params(self, k: int = 0) -> torch.Tensor:
        return self.params[k * self.n_obj:(k + 1) * self.n_obj]

    def view_params_flat(self) -> torch.Tensor:
        return self.params[self.n_all_obj:]

    def set_noise(self, eps, noise, timesteps):
        """host-supplied randomness (parity tests: identical values for the oracle and the GPU)."""
        self.eps.copy_(eps)
        self.noise.copy_(noise)
        self.timesteps.copy_(timesteps)

    def set_lr(self, lr: float):
        st = self.stager  # (a scalar write `hyper[0] = lr` is a blocking upload too)
        st.begin()
        st.upload("lr", self.hyper[0:1], torch.tensor([lr], dtype=torch.float32))
        st.end()

    # ------------------------------------------------------------------ the step
    def forward_backward(self, accumulate: bool = False, cached: bool = False):
        """cached: the batch's VAE moments come out of the moment cache (no encoder launches); otherwise the encoder runs
        and, with a cache, its moments are stored at the batch's image slots."""
        B, Lc, hw = self.B, self.cfg.vae.latent_channels, self.h * self.w
        self.text.accumulate_grads = accumulate
        if self.device_rng:
            ops.rng_advance(self.rng_state)
            ops.rng_fill_randint(self.timesteps, self.cfg.ddpm.num_train_timesteps, self.rng_state, 0)
            ops.rng_fill_normal(self.eps, self.rng_state, 1)
            ops.rng_fill_normal(self.noise, self.rng_state, 2)
        # overlap=True forks: the 16 mapper+CLIP passes (many small launches) run beside the VAE encoder (few large ones) on
        # a second stream; each schedule owns its split-K scratch, so they never alias.  OFF by default since round 6: both
        # sides fill the chip, and the two cross-stream edges of the captured graph cost more than the overlap returns
        # (same box, alternating processes: 39.96 / 39.98 / 40.05 steps/s without the fork, 39.08 / 39.75 / 39.79 with it;
        # profiles/r06_halo_persist_ab.txt) — the step is ONE linear chain of launches
        main = torch.cuda.current_stream()
        if self.overlap:
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                self.text.forward()
                self.unet.forward_pre()  # time embedding + the 32 context K/V projections
        if cached:
            torch.index_select(self.mcache, 0, self.img_idx, out=self.vae.moments.view(B, hw, 2 * Lc))
        else:
            self.vae.forward()
            if self.n_cache:
                self.mcache.index_copy_(0, self.img_idx, self.vae.moments.view(B, hw, 2 * Lc))
        ops.sample_add_noise(self.vae.moments, self.eps, self.noise, self.timesteps, self.ac,
                             self.cfg.vae.scaling_factor, self.cfg.ddpm.prediction_type == "v_prediction", self.latents,
                             self.unet.x_in, self.target, B, Lc, hw)
        if self.overlap:
            main.wait_stream(self.side)  # join
        else:
            self.text.forward()
            self.unet.forward_pre()
        self.unet.forward_main()
        self.loss_sum.zero_()
        ops.mse_loss_grad(self.unet.pred, self.target, self.unet.dpred, self.loss_sum, self.scaler, B, Lc, hw)
        if self.need_backward:
            self.unet.backward()
            self.text.backward()

    def optimizer_step(self):
        a = (self.hyper, self.scaler, self.opt_step, self.growth_interval)
        if self.n_objects == 1:
            ops.adamw_flat(self.params, self.grads, self.exp_avg, self.exp_avg_sq, *a)
            return
        # several buckets, one optimizer step: check all, apply all, then GradScaler.update once
        no = self.n_all_obj
        obj = (self.params[:no], self.grads[:no], self.exp_avg[:no], self.exp_avg_sq[:no], self.n_obj, self.n_objects,
               self.seg_step, self.obj_slot)
        view = (self.params[no:], self.grads[no:], self.exp_avg[no:], self.exp_avg_sq[no:])
        has_view = self.params.numel() > no
        if has_view:
            ops.adamw_flat(*view, *a, phases=ops.OPT_CHECK)
        ops.adamw_segments(*obj, *a, phases=ops.OPT_CHECK)
        if has_view:
            ops.adamw_flat(*view, *a, phases=ops.OPT_APPLY)
        ops.adamw_segments(*obj, *a, phases=ops.OPT_APPLY | ops.OPT_FINISH)

    def all_reduce(self):
        """the one exchange step of data-parallel training: sum the mapper gradients over ranks
        (the 1/world_size is folded into AdamW).  With several object mappers every rank trains the
        same scene per step (same scene-sampler seed), so only that segment and the view mapper move."""
        if self.world_size > 1:
            from ..parallel import all_reduce_plan_, reduce_plan
            plan = reduce_plan(self.n_obj, self.n_objects, self.active_object, self.grads.numel())
            if len(plan) > 1 and getattr(self, "_reduce_stage", None) is None:  # scene segment + view mapper, packed
                self._reduce_stage = torch.empty(sum(b - a for a, b in plan), dtype=self.grads.dtype, device=self.grads.device)
            self.last_reduce_bytes = all_reduce_plan_(self.grads, plan, getattr(self, "_reduce_stage", None),
                                                      comm=self.exchange)

    def _exchange_capturable(self) -> bool:
        """the all-reduce may sit INSIDE a graph: a stream-ordered library communicator, and a plan that does not depend on
        host state (several object mappers pack the active scene's segment, chosen per step on the host)"""
        return self.world_size > 1 and self.exchange is not None and self.n_objects == 1

    def _use_cache(self) -> bool:
        """decided on the host per micro-batch: does the batch consist of cached images only"""
        return bool(self.n_cache) and self._batch_cached

    def _mark_cached(self, used_cache: bool):
        """after the micro-step is enqueued: a batch that ran the encoder has its images' moments in the cache (stream order
        makes them visible to every later step).  Marking BEFORE the launch would leave stale slots behind a step that
        failed to enqueue."""
        if self.n_cache and not used_cache:
            self._cached_images.update(self._batch_images)
            self._batch_cached = bool(self._batch_images)  # the same batch again is served from the cache

    def reset_moment_cache(self, image_idx=None):
        """forget cached moments: all of them, or the listed dataset indices — for callers whose pixels behind an index
        change (the cache is only as deterministic as the dataset the CALLER vouched for)"""
        if not self.n_cache:
            return
        if image_idx is None:
            self._cached_images.clear()
        else:
            self._cached_images.difference_update(int(i) for i in image_idx)
        self._batch_cached = all(i in self._cached_images for i in self._batch_images) and bool(self._batch_images)

    def step_eager(self):
        """one micro-step; the optimizer runs after every `grad_accum`-th micro-step."""
        cached = self._use_cache()
        self.forward_backward(accumulate=self.micro > 0, cached=cached)
        self._mark_cached(cached)
        self.micro += 1
        if self.micro == self.grad_accum:
            self.micro = 0
            self.all_reduce()
            self.optimizer_step()
            return True
        return False

    # ------------------------------------------------------------------ hipGraph capture
    def capture(self):
        """Capture the step into ONE hipGraph: forward + backward, (world > 1 on RCCL: the all-reduce of the flat gradient
        bucket as one collective node,) fused AdamW — the launch list of N GPUs is the one-GPU list plus that node.
        Fallbacks keep the older shape [forward+backward] -> host-issued all-reduce -> [optimizer]: the gloo backend,
        several object mappers (host-dependent exchange plan), and a communicator that refuses capture."""
        import os
        want = self._exchange_capturable()
        err = None
        try:
            self._capture(want)
        except RuntimeError as e:
            if not want:
                raise
            err = e
        if not want:
            return
        # the route is a COLLECTIVE decision: one rank replaying [graph A -> host all-reduce on torch's communicator ->
        # graph B] beside ranks whose collective sits in their graph on the library's communicator would never pair up
        from ..parallel import all_agree
        if all_agree(err is None):
            return
        if os.environ.get("VNETI_REQUIRE_ONE_GRAPH", "0") == "1":
            raise RuntimeError("VNETI_REQUIRE_ONE_GRAPH=1: the RCCL all-reduce could not be captured into the step's graph "
                               f"on {'this' if err else 'another'} rank" + (f" ({err})" if err else ""))
        import warnings
        warnings.warn(f"capturing the RCCL all-reduce failed on {'this' if err else 'another'} rank"
                      + (f" ({err})" if err else "") + "; every rank runs the exchange between two graphs")
        torch.cuda.synchronize()
        self.graph_a = self.graph_b = self.graph_acc = self.graph_a_c = self.graph_acc_c = None
        self._capture(False)

    def _capture(self, exchange_in_graph: bool):
        # the warm-up launches below are real steps: snapshot the trainable / RNG state and put it back — ALSO when the
        # capture raises (the fallback re-captures from the same state, not from one optimizer step later)
        saved = (self.params, self.exp_avg, self.exp_avg_sq, self.opt_step, self.scaler, self.rng_state, self.seg_step)
        state = [t.clone() for t in saved]
        try:
            self._capture_graphs(exchange_in_graph)
        finally:
            torch.cuda.synchronize()
            for dst, src in zip(saved, state):
                dst.copy_(src)
            self.micro = 0

    def _capture_graphs(self, exchange_in_graph: bool):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.grad_accum):
                self.step_eager()  # warm-up on the side stream (also primes RCCL)
            torch.cuda.synchronize()
            fused_opt = (self.world_size == 1 or exchange_in_graph) and self.grad_accum == 1
            self.exchange_in_graph = exchange_in_graph
            self.graph_a = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_a, stream=s):
                self.forward_backward(accumulate=False)
                if fused_opt:
                    self.all_reduce()  # no-op at world 1; one captured collective node otherwise
                    self.optimizer_step()
            if self.grad_accum > 1:
                self.graph_acc = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_acc, stream=s):
                    self.forward_backward(accumulate=True)
            if self.n_cache:  # the same steps with the moments read from the cache instead of the encoder
                self.graph_a_c = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_a_c, stream=s):
                    self.forward_backward(accumulate=False, cached=True)
                    if fused_opt:
                        self.all_reduce()
                        self.optimizer_step()
                if self.grad_accum > 1:
                    self.graph_acc_c = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph_acc_c, stream=s):
                        self.forward_backward(accumulate=True, cached=True)
            if not fused_opt:
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, stream=s):
                    if exchange_in_graph:
                        self.all_reduce()
                    self.optimizer_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

    def step(self) -> bool:
        """one micro-step (graph replay when captured); returns True when the optimizer stepped."""
        if self.graph_a is None:
            return self.step_eager()
        if self.exchange_in_graph and getattr(self.exchange, "closed", False):
            raise RuntimeError("the step's graph holds a collective on an RCCL communicator that has been closed (process "
                               "group re-initialised?): rebuild the engine")
        cached = self._use_cache()
        if cached:
            (self.graph_a_c if self.micro == 0 else self.graph_acc_c).replay()
        else:
            (self.graph_a if self.micro == 0 else self.graph_acc).replay()
        self._mark_cached(cached)
        self.micro += 1
        if self.micro < self.grad_accum:
            return False
        self.micro = 0
        if self.graph_b is not None:
            if not self.exchange_in_graph:
                self.all_reduce()
            self.graph_b.replay()
        if self.exchange_in_graph:
            from .. import parallel
            parallel.COLLECTIVE_CALLS += 1  # the replayed collective node
        return True

    def loss(self) -> float:
        """mean squared error of the last step (forces a device sync — call sparingly)."""
        return float(self.loss_sum.item()) / self.n_loss

    # ------------------------------------------------------------------ introspection
    def launches(self) -> List:
        out = list(self.vae.fwd) + list(self.text.fwd) + list(self.unet.fwd_pre) + list(self.unet.fwd)
        if self.need_backward:
            out += list(self.unet.bwd) + list(self.text.bwd)
        return out

    def memory_bytes(self) -> int:
        return self.unet.bytes + self.vae.bytes + self.text.bytes
