"""Auto-decoder training driver: the API and on-disk artefacts of the reference's train_deep_sdf.py
(main_function :255-581, save_*/load_* :96-218, LR schedules :23-93) around the fused MI355X step.

What differs from the reference, by design:
  * the per-step body (:481-545) is ONE call sequence into libdsdf_hip.so (Engine.train_forward_backward /
    grad_norm / adam_step) on device-resident state; no per-step H2D copy, no per-chunk .item() sync (losses are
    collected in a device buffer and read once per epoch);
  * samples come from a DeviceSampleCache instead of DataLoader workers;
  * multi-GPU = one process per GPU (torchrun), scenes sharded by owner rank, one RCCL all-reduce of the decoder
    gradient arena per step (dist.py) instead of nn.DataParallel.
Checkpoints (ModelParameters/, OptimizerParameters/, LatentCodes/, Logs.pth) keep the reference's key layout,
including the ``module.`` prefix nn.DataParallel gave every decoder key, so either trainer can resume the other's run.
"""
import datetime
import json
import logging
import math
import os
import signal
import sys
import time

import torch

from . import dist
from . import workspace as ws
from .data import get_instance_filenames, make_sample_cache
from .engine import make_segments


# ---- learning-rate schedules (train_deep_sdf.py:23-93) ------------------------------------------------------
class LearningRateSchedule:
    def get_learning_rate(self, epoch):
        raise NotImplementedError


class ConstantLearningRateSchedule(LearningRateSchedule):
    def __init__(self, value):
        self.value = value

    def get_learning_rate(self, epoch):
        return self.value


class StepLearningRateSchedule(LearningRateSchedule):
    def __init__(self, initial, interval, factor):
        self.initial, self.interval, self.factor = initial, interval, factor

    def get_learning_rate(self, epoch):
        return self.initial * (self.factor ** (epoch // self.interval))


class WarmupLearningRateSchedule(LearningRateSchedule):
    def __init__(self, initial, warmed_up, length):
        self.initial, self.warmed_up, self.length = initial, warmed_up, length

    def get_learning_rate(self, epoch):
        if epoch > self.length:
            return self.warmed_up
        return self.initial + (self.warmed_up - self.initial) * epoch / self.length


def get_learning_rate_schedules(specs):
    schedules = []
    for s in specs["LearningRateSchedule"]:
        if s["Type"] == "Step":
            schedules.append(StepLearningRateSchedule(s["Initial"], s["Interval"], s["Factor"]))
        elif s["Type"] == "Warmup":
            schedules.append(WarmupLearningRateSchedule(s["Initial"], s["Final"], s["Length"]))
        elif s["Type"] == "Constant":
            schedules.append(ConstantLearningRateSchedule(s["Value"]))
        else:
            raise Exception('no known learning rate schedule of type "{}"'.format(s["Type"]))
    return schedules


def get_spec_with_default(specs, key, default):
    return specs[key] if key in specs else default


# ---- optimizer state in torch.optim.Adam's own state_dict format ---------------------------------------------
class AdamStateBridge:
    """Exposes the fused Adam's flat moment arenas as a real ``torch.optim.Adam`` (never stepped) so that
    ``state_dict()`` / ``load_state_dict()`` produce and accept exactly the reference's
    ``optimizer_state_dict`` (2 param groups: decoder tensors in named_parameters order, then the latent table;
    per-tensor ``step``, ``exp_avg``, ``exp_avg_sq``; train_deep_sdf.py:106-131,400-411)."""

    def __init__(self, decoder, engine, lat_param, lat_m, lat_v, lr0, lr1):
        self.decoder, self.engine = decoder, engine
        self.lat_param, self.lat_m, self.lat_v = lat_param, lat_m, lat_v
        self.opt = torch.optim.Adam([{"params": list(decoder.parameters()), "lr": lr0},
                                     {"params": [lat_param], "lr": lr1}])

    def set_lrs(self, lr0, lr1):
        self.opt.param_groups[0]["lr"], self.opt.param_groups[1]["lr"] = lr0, lr1

    def _pairs(self):
        eng = self.engine
        for par, info in zip(self.decoder.parameters(), eng.spec.params):
            yield par, eng.view(eng.exp_avg, info), eng.view(eng.exp_avg_sq, info)
        yield self.lat_param, self.lat_m, self.lat_v

    def state_dict(self):
        if self.engine.step > 0:
            for par, m, v in self._pairs():
                self.opt.state[par] = {"step": torch.tensor(float(self.engine.step)), "exp_avg": m, "exp_avg_sq": v}
        return self.opt.state_dict()

    def load_state_dict(self, sd):
        self.opt.load_state_dict(sd)
        step = 0
        for par, m, v in self._pairs():
            st = self.opt.state.get(par)
            if st:
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                step = max(step, int(float(st["step"])))
        self.opt.state.clear()
        self.engine.step = step


# ---- checkpoint io (train_deep_sdf.py:96-218) -----------------------------------------------------------------
def save_model(experiment_directory, filename, decoder, epoch):
    sd = {"module." + k: v.detach().cpu() for k, v in decoder.state_dict().items()}   # nn.DataParallel's prefix (:353)
    torch.save({"epoch": epoch, "model_state_dict": sd},
               os.path.join(ws.get_model_params_dir(experiment_directory, True), filename))


def save_optimizer(experiment_directory, filename, optimizer, epoch):
    torch.save({"epoch": epoch, "optimizer_state_dict": optimizer.state_dict()},
               os.path.join(ws.get_optimizer_params_dir(experiment_directory, True), filename))


def load_optimizer(experiment_directory, filename, optimizer):
    full_filename = os.path.join(ws.get_optimizer_params_dir(experiment_directory), filename)
    if not os.path.isfile(full_filename):
        raise Exception('optimizer state dict "{}" does not exist'.format(full_filename))
    data = torch.load(full_filename, map_location="cpu", weights_only=True)
    optimizer.load_state_dict(data["optimizer_state_dict"])
    return data["epoch"]


def save_latent_vectors(experiment_directory, filename, latent_weight, epoch):
    torch.save({"epoch": epoch, "latent_codes": {"weight": latent_weight.detach().cpu()}},
               os.path.join(ws.get_latent_codes_dir(experiment_directory, True), filename))


def load_latent_vectors(experiment_directory, filename, latent_weight):
    full_filename = os.path.join(ws.get_latent_codes_dir(experiment_directory), filename)
    if not os.path.isfile(full_filename):
        raise Exception('latent state file "{}" does not exist'.format(full_filename))
    data = torch.load(full_filename, map_location="cpu", weights_only=True)
    codes = data["latent_codes"]
    if isinstance(codes, torch.Tensor):    # legacy upstream layout [num, 1, L]
        if not latent_weight.shape[0] == codes.size()[0]:
            raise Exception("num latent codes mismatched: {} vs {}".format(latent_weight.shape[0], codes.size()[0]))
        if not latent_weight.shape[1] == codes.size()[2]:
            raise Exception("latent code dimensionality mismatch")
        latent_weight.copy_(codes.reshape(codes.size()[0], -1))
    else:
        w = codes["weight"]
        if tuple(w.shape) != tuple(latent_weight.shape):
            raise Exception("num latent codes mismatched: {} vs {}".format(tuple(latent_weight.shape), tuple(w.shape)))
        latent_weight.copy_(w)
    return data["epoch"]


def save_logs(experiment_directory, loss_log, lr_log, timing_log, lat_mag_log, param_mag_log, epoch):
    torch.save({"epoch": epoch, "loss": loss_log, "learning_rate": lr_log, "timing": timing_log,
                "latent_magnitude": lat_mag_log, "param_magnitude": param_mag_log},
               os.path.join(experiment_directory, ws.logs_filename))


def load_logs(experiment_directory):
    full_filename = os.path.join(experiment_directory, ws.logs_filename)
    if not os.path.isfile(full_filename):
        raise Exception('log file "{}" does not exist'.format(full_filename))
    d = torch.load(full_filename, map_location="cpu", weights_only=True)
    return d["loss"], d["learning_rate"], d["timing"], d["latent_magnitude"], d["param_magnitude"], d["epoch"]


def clip_logs(loss_log, lr_log, timing_log, lat_mag_log, param_mag_log, epoch):
    iters_per_epoch = len(loss_log) // len(lr_log)
    for n in param_mag_log:
        param_mag_log[n] = param_mag_log[n][:epoch]
    return (loss_log[:iters_per_epoch * epoch], lr_log[:epoch], timing_log[:epoch], lat_mag_log[:epoch], param_mag_log)


def get_mean_latent_vector_magnitude(latent_weight):
    return torch.mean(torch.norm(latent_weight.detach(), dim=1)).cpu()


class EpochStats:
    """The per-epoch log values of train_deep_sdf.py:548-590 (step losses, mean latent-vector magnitude, parameter
    magnitudes) without draining the GPU queue: they are reduced on the device, leave it in ONE asynchronous copy into
    pinned memory, and are consumed one epoch later (`flush` before anything is saved)."""

    def __init__(self, device, steps, decoder):
        self.names = [n[7:] if n.startswith("module.") else n for n, _ in decoder.named_parameters()]
        self.params = [p for _, p in decoder.named_parameters()]
        self.steps = steps
        n = steps + 1 + len(self.names)
        self.dev = torch.empty(2, n, device=device)
        self.host = torch.empty(2, n).pin_memory() if torch.device(device).type == "cuda" else torch.empty(2, n)
        self.events = [torch.cuda.Event(), torch.cuda.Event()]
        self.pending = None                                  # slot of the epoch not consumed yet

    def push(self, epoch, loss_buf, latents, lat_mag=None):
        """lat_mag: the mean latent magnitude when the caller already has it (world > 1: the mean over ALL ranks' rows)."""
        slot = epoch & 1
        d = self.dev[slot]
        d[:self.steps].copy_(loss_buf[:self.steps])
        d[self.steps] = torch.mean(torch.norm(latents.detach(), dim=1)) if lat_mag is None else lat_mag
        d[self.steps + 1:].copy_(torch.stack(torch._foreach_norm([p.data for p in self.params])))
        self.host[slot].copy_(d, non_blocking=True)
        self.events[slot].record()
        self.pending = slot

    def pop(self, loss_log, lat_mag_log, param_mag_log):
        """Append the pending epoch (waits for its copy; a no-op when nothing is pending)."""
        if self.pending is None:
            return
        slot, self.pending = self.pending, None
        self.events[slot].synchronize()
        h = self.host[slot]
        loss_log.extend(h[:self.steps].tolist())
        lat_mag_log.append(h[self.steps].clone())
        for name, v in zip(self.names, h[self.steps + 1:].tolist()):
            param_mag_log.setdefault(name, []).append(v)


# ---- the fused step ---------------------------------------------------------------------------------------------
class FusedTrainStep:
    """train_deep_sdf.py:483-545 for one batch, on the device: chunking (--batch_split), renorm + gather + forward +
    loss + backward per chunk, optional clip, [all-reduce], Adam on both groups."""

    def __init__(self, engine, latents, *, clamp_dist, code_reg, code_reg_lambda, code_bound, grad_clip, seed=0):
        self.eng, self.lat = engine, latents
        self.dlat = torch.zeros_like(latents)
        self.lat_m, self.lat_v = torch.zeros_like(latents), torch.zeros_like(latents)
        self.clamp_dist, self.code_reg, self.lam = clamp_dist, code_reg, code_reg_lambda
        self.code_bound, self.grad_clip, self.seed = code_bound, grad_clip, seed
        # DSDF_AR_BUCKETS=K (2..8, data-parallel steps only): the decoder gradient is exchanged in K buckets, last layers first --
        # bucket b's all-reduce runs under the weight-gradient launches of the buckets below it.  Default 1:
        # ONE all-reduce.  Decided once, here: a net the fused kernels do not cover exchanges its gradient in one piece on every
        # rank alike (ranks deciding differently would issue mismatched collectives).
        from ._lib import MAX_BUCKETS
        want = os.environ.get("DSDF_AR_BUCKETS", "1")
        if not want.isdigit() or not 1 <= int(want) <= MAX_BUCKETS:
            raise ValueError("DSDF_AR_BUCKETS must be an integer in [1, {}], got {!r}".format(MAX_BUCKETS, want))
        self.ar_buckets = int(want) if int(want) > 1 and engine.dw_phase_supported() else 1
        self._bucket_off = engine.grad_buckets(self.ar_buckets)[1] if self.ar_buckets > 1 else None

    def __call__(self, scene_rows, samples_per_scene, xyz, sdf_gt, epoch, lr_decoder, lr_latent, batch_split=1,
                 n_norm=None, under_allreduce=None, loss_out=None):
        """scene_rows [B] (rows of self.lat), xyz [B*S, G], sdf_gt [B*S]; returns nothing (loss in eng.loss).
        under_allreduce: optional callable that enqueues work independent of this step's decoder gradient (e.g. the next
        batch's sampling); it runs exactly once per call, under the gradient all-reduce when there is one.
        loss_out: 1-element fp32 device tensor that receives the step's loss instead of eng.loss (the trainer passes the
        step's slot of its per-epoch loss buffer: the kernels write it in place, no copy launch per step)."""
        N = xyz.shape[0]
        n_norm = N if n_norm is None else n_norm
        uniform = 0
        reg = self.lam * min(1, epoch / 100) if self.code_reg else 0.0
        if batch_split == 1:
            seg_scene = scene_rows
            ck = (N, samples_per_scene)
            if getattr(self, "_seg_off_key", None) != ck:      # the same every step: built once
                self._seg_off = torch.arange(0, N + 1, samples_per_scene, dtype=torch.int64, device=xyz.device)
                self._seg_off_key = ck
            seg_off = self._seg_off
            chunks = [(seg_scene, seg_off, xyz, sdf_gt)]
            uniform = samples_per_scene
        else:
            idx = scene_rows.repeat_interleave(samples_per_scene)
            if N % batch_split == 0 and (N // batch_split) % samples_per_scene == 0:
                uniform = samples_per_scene      # every torch.chunk piece holds whole scenes: segment mode per chunk
            chunks = []
            for ic, xc, gc in zip(torch.chunk(idx, batch_split), torch.chunk(xyz, batch_split),
                                  torch.chunk(sdf_gt, batch_split)):
                sc, so = make_segments(ic)
                chunks.append((sc, so, xc.contiguous(), gc.contiguous()))
        force_dp = os.environ.get("DSDF_FORCE_DP_PATH") == "1"     # measurement knob: the world > 1 call sequence on one process
        if batch_split == 1 and self.grad_clip is None and not dist.is_multi() and not force_dp:
            # single-GPU fast path: one library call, decoder Adam folded into the finalize pass
            sc, so, xc, gc = chunks[0]
            self.eng.train_step(self.lat, self.dlat, self.lat_m, self.lat_v, sc, so, xc, gc, n_norm=n_norm,
                                clamp_dist=self.clamp_dist, reg_coef=reg, code_bound=self.code_bound, lr_decoder=lr_decoder,
                                lr_latent=lr_latent, training=True, seed=self.seed, seg_len=uniform, loss_out=loss_out)
            if under_allreduce is not None:
                under_allreduce()
            return
        def fb(ci, sc, so, xc, gc, row0, phase=0):
            self.eng.train_forward_backward(self.lat, self.dlat, sc, so, xc, gc, n_norm=n_norm, clamp_dist=self.clamp_dist,
                                            reg_coef=reg, code_bound=self.code_bound, training=True, seed=self.seed,
                                            row_offset=row0, accumulate=ci > 0, seg_len=uniform, loss_out=loss_out, dw_phase=phase,
                                            dw_buckets=self.ar_buckets if phase else 0)

        # Data parallel (train_deep_sdf.py:353 replaced): a sum all-reduce of the decoder-gradient
        # arena, issued asynchronously (RCCL runs it on its own stream, ordered after the finalize launch that wrote the
        # arena).  Everything that does not need the reduced gradient is enqueued on the compute stream meanwhile: the Adam
        # update of this rank's latent rows (their gradient is complete and private to the owner rank) and `under_allreduce`
        # (the trainer passes the NEXT batch's sampling kernel).  The decoder's Adam + weight re-materialisation wait for it.
        works = None
        if self.ar_buckets > 1 and len(chunks) == 1 and (dist.is_multi() or force_dp):
            # K buckets, last layers first: phase 1 leaves bucket 0's gradients in the arena -> their all-reduce starts and runs
            # under phase 2 (the next bucket's weight-gradient launch + finalize) -> ... -> the last bucket follows
            works, off = [], self._bucket_off
            for p in range(1, self.ar_buckets + 1):
                fb(0, *chunks[0], 0, phase=p)
                if off[p] < off[p - 1]:          # (a net with fewer layers than buckets leaves trailing buckets empty)
                    works.append(dist.allreduce_sum_async(self.eng.grads[off[p]:off[p - 1]]))
        if works is None:
            row0 = 0
            for ci, (sc, so, xc, gc) in enumerate(chunks):
                fb(ci, sc, so, xc, gc, row0)
                row0 += xc.shape[0]
            works = [dist.allreduce_sum_async(self.eng.grads)]
        works = [w for w in works if w is not None]
        split_adam = bool(works) or (force_dp and self.grad_clip is None)
        if split_adam:
            self.eng.adam_latents(self.lat, self.dlat, self.lat_m, self.lat_v, lr_latent)
            if under_allreduce is not None:
                under_allreduce()
            for w in works:
                w.wait()                      # nccl: the compute stream waits for RCCL's stream (no host block)
        if self.grad_clip is not None:
            self.eng.grad_norm(self.grad_clip)
        if split_adam:
            self.eng.adam_step(None, None, None, None, lr_decoder, lr_latent, clip=self.grad_clip is not None)
        else:
            self.eng.adam_step(self.lat, self.dlat, self.lat_m, self.lat_v, lr_decoder, lr_latent, clip=self.grad_clip is not None)
            if under_allreduce is not None:
                under_allreduce()


# ---- driver ---------------------------------------------------------------------------------------------------------
def main_function(experiment_directory, continue_from, batch_split):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s", datefmt="%H:%M:%S")
    logging.debug("running " + experiment_directory)
    specs = ws.load_experiment_specifications(experiment_directory)
    logging.info("Experiment description: \n" + specs["Description"])

    data_source, train_split_file = specs["DataSource"], specs["TrainSplit"]
    _ = specs["ReconstructionSplit"]       # read unconditionally by the reference (:268)
    arch = __import__("deep_sdf.networks." + specs["NetworkArch"], fromlist=["Decoder"])
    latent_size = specs["CodeLength"]
    checkpoints = list(range(specs["SnapshotFrequency"], specs["NumEpochs"] + 1, specs["SnapshotFrequency"]))
    checkpoints += list(specs["AdditionalSnapshots"])
    checkpoints.sort()
    lr_schedules = get_learning_rate_schedules(specs)
    grad_clip = get_spec_with_default(specs, "GradientClipNorm", None)

    def signal_handler(sig, frame):
        logging.info("Stopping early...")
        sys.exit(0)

    signal.signal(signal.SIGINT, signal_handler)

    num_samp_per_scene, scene_per_batch = specs["SamplesPerScene"], specs["ScenesPerBatch"]
    clamp_dist = specs["ClampingDistance"]
    do_code_regularization = get_spec_with_default(specs, "CodeRegularization", True)
    code_reg_lambda = get_spec_with_default(specs, "CodeRegularizationLambda", 1e-4)
    code_bound = get_spec_with_default(specs, "CodeBound", None)
    num_epochs = specs["NumEpochs"]
    log_frequency = get_spec_with_default(specs, "LogFrequency", 10)

    rank, local, world = dist.init()
    multi = dist.is_multi()       # a process group drives the step: world > 1, or the one-rank rehearsal group (DSDF_DIST_FORCE_GROUP=1)
    if not torch.cuda.is_available():
        raise RuntimeError("train_deep_sdf (deepsdf_amd) needs an AMD GPU: the HIP training step has no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    decoder = arch.Decoder(latent_size, **specs["NetworkSpecs"]).to(device)
    if not hasattr(decoder, "engine"):
        raise RuntimeError("NetworkArch '{}' does not resolve to the HIP decoder".format(specs["NetworkArch"]))
    geom_dimension = decoder.geom_dimension
    if multi:      # identical replicas: rank 0's initialisation wins
        torch.distributed.broadcast(decoder._arena, src=0)
    eng = decoder.engine()
    logging.info("training with {} GPU(s), one process each".format(world))

    with open(train_split_file, "r") as f:
        train_split = json.load(f)
    npzfiles = get_instance_filenames(data_source, train_split)
    num_scenes = len(npzfiles)
    logging.info("There are {} scenes".format(num_scenes))
    lo, hi = dist.owned_scenes(num_scenes, rank, world)
    # resident in HBM, or -- datasets beyond the budget -- staged through pinned host memory (same batches either way)
    cache = make_sample_cache(data_source, npzfiles[lo:hi], geom_dimension, device, max_batch_scenes=scene_per_batch)

    # latent table: Embedding(num_scenes, L) ~ N(0, CodeInitStdDev/sqrt(L)) (:385-390); every rank draws the full table
    # from the same seed and keeps its own rows
    full = torch.empty(num_scenes, latent_size)
    torch.nn.init.normal_(full, 0.0, get_spec_with_default(specs, "CodeInitStdDev", 1.0) / math.sqrt(latent_size))
    if multi:
        objs = [full if rank == 0 else None]
        torch.distributed.broadcast_object_list(objs, src=0)
        full = objs[0]
    lat = full[lo:hi].to(device).contiguous()
    logging.debug("initialized with mean magnitude {}".format(get_mean_latent_vector_magnitude(lat)))

    fused = FusedTrainStep(eng, lat, clamp_dist=clamp_dist, code_reg=do_code_regularization,
                           code_reg_lambda=code_reg_lambda, code_bound=code_bound, grad_clip=grad_clip,
                           seed=int(torch.initial_seed() & 0x7FFFFFFF) + rank)
    lat_param = torch.nn.Parameter(lat, requires_grad=True)        # shares storage with `lat`
    optimizer_all = AdamStateBridge(decoder, eng, lat_param, fused.lat_m, fused.lat_v,
                                    lr_schedules[0].get_learning_rate(0), lr_schedules[1].get_learning_rate(0))

    def gather_latents(t):
        if not multi:
            return t
        parts = [None] * world
        torch.distributed.all_gather_object(parts, t.detach().cpu())
        return torch.cat(parts, 0)

    def save_all(name, epoch):
        full_lat, full_m, full_v = gather_latents(lat), gather_latents(fused.lat_m), gather_latents(fused.lat_v)
        if multi:      # rank 0's decoder is what gets saved: it must BE every rank's decoder (bit for bit)
            for what, arena in (("parameters", eng.params), ("exp_avg", eng.exp_avg), ("exp_avg_sq", eng.exp_avg_sq)):
                if not dist.replicas_identical(arena):
                    raise RuntimeError("data-parallel replicas diverged: decoder {} differ between ranks at epoch {}".format(
                        what, epoch))
            logging.info("epoch {}: decoder replicas bit-identical on {} ranks".format(epoch, world))
        if rank != 0:
            return
        save_model(experiment_directory, name, decoder, epoch)
        if not multi:
            save_optimizer(experiment_directory, name, optimizer_all, epoch)
        else:   # write the optimizer state of the FULL latent table
            tmp = AdamStateBridge(decoder, eng, torch.nn.Parameter(full_lat.to(device)), full_m.to(device), full_v.to(device),
                                  optimizer_all.opt.param_groups[0]["lr"], optimizer_all.opt.param_groups[1]["lr"])
            save_optimizer(experiment_directory, name, tmp, epoch)
        save_latent_vectors(experiment_directory, name, full_lat, epoch)

    loss_log, lr_log, lat_mag_log, timing_log, param_mag_log = [], [], [], [], {}
    start_epoch = 1
    if continue_from is not None:
        logging.info('continuing from "{}"'.format(continue_from))
        full_lat = torch.empty(num_scenes, latent_size)
        lat_epoch = load_latent_vectors(experiment_directory, continue_from + ".pth", full_lat)
        lat.copy_(full_lat[lo:hi])
        model_epoch = ws.load_model_parameters(experiment_directory, continue_from, torch.nn.DataParallel(decoder))
        eng.weights_dirty = True
        if not multi:
            optimizer_epoch = load_optimizer(experiment_directory, continue_from + ".pth", optimizer_all)
        else:
            fm, fv = torch.zeros(num_scenes, latent_size, device=device), torch.zeros(num_scenes, latent_size, device=device)
            tmp = AdamStateBridge(decoder, eng, torch.nn.Parameter(full_lat.to(device)), fm, fv, 0.0, 0.0)
            optimizer_epoch = load_optimizer(experiment_directory, continue_from + ".pth", tmp)
            fused.lat_m.copy_(fm[lo:hi]); fused.lat_v.copy_(fv[lo:hi])
        loss_log, lr_log, timing_log, lat_mag_log, param_mag_log, log_epoch = load_logs(experiment_directory)
        if not log_epoch == model_epoch:
            loss_log, lr_log, timing_log, lat_mag_log, param_mag_log = clip_logs(
                loss_log, lr_log, timing_log, lat_mag_log, param_mag_log, model_epoch)
        if not (model_epoch == optimizer_epoch and model_epoch == lat_epoch):
            raise RuntimeError("epoch mismatch: {} vs {} vs {} vs {}".format(model_epoch, optimizer_epoch, lat_epoch, log_epoch))
        start_epoch = model_epoch + 1
        logging.debug("loaded")

    logging.info("starting from epoch {}".format(start_epoch))
    logging.info("Number of decoder parameters: {}".format(sum(p.data.nelement() for p in decoder.parameters())))
    logging.info("Number of shape code parameters: {} (# codes {}, code dim {})".format(
        num_scenes * latent_size, num_scenes, latent_size))

    # Batch semantics at world > 1.  The reference wraps the decoder in nn.DataParallel (train_deep_sdf.py:353), which
    # SPLITS the one ScenesPerBatch batch over the GPUs (:514): the same specs.json trains with the same global batch and
    # learning rate on any number of GPUs.  Default here = the same: every rank takes ScenesPerBatch / world scenes per
    # step.  DSDF_SCENES_PER_BATCH_PER_RANK=1 keeps ScenesPerBatch scenes PER RANK instead (weak scaling: the global batch
    # grows with the world size and the learning rate is NOT rescaled -- a different optimisation problem, opt-in only).
    n_local = hi - lo
    per_rank = os.environ.get("DSDF_SCENES_PER_BATCH_PER_RANK") == "1"
    if multi and not per_rank:
        if scene_per_batch % world != 0:
            raise RuntimeError("ScenesPerBatch {} is not divisible by the {} data-parallel ranks (nn.DataParallel would split "
                               "the batch unevenly; set DSDF_SCENES_PER_BATCH_PER_RANK=1 for a per-rank batch)".format(
                                   scene_per_batch, world))
        local_batch = scene_per_batch // world
    else:
        local_batch = scene_per_batch
    steps_per_epoch = n_local // local_batch                 # drop_last=True (:374)
    if multi:                                            # every rank must take the same number of steps
        t = torch.tensor([steps_per_epoch], device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
        steps_per_epoch = int(t.item())
    if steps_per_epoch == 0:
        raise RuntimeError("no full batch: {} scenes on the smallest of {} rank(s) but {} scenes per batch and rank "
                           "(the reference's DataLoader(drop_last=True) would train nothing either)".format(
                               n_local, world, local_batch))
    loss_buf = torch.zeros(steps_per_epoch, device=device)
    stats = EpochStats(device, steps_per_epoch, decoder)
    order_pinned = [torch.empty(n_local, dtype=torch.int64).pin_memory() for _ in range(2)]
    order_device = [torch.empty(n_local, dtype=torch.int64, device=device) for _ in range(2)]
    gen = torch.Generator(device=device)
    gen.manual_seed(int(torch.initial_seed() & 0x7FFFFFFF) + 7919 * rank)
    n_norm = local_batch * num_samp_per_scene * world        # loss normaliser = GLOBAL points per step (:519)
    start_train = time.time()
    for epoch in range(start_epoch, num_epochs + 1):
        start = time.time()
        decoder.train()
        lr0, lr1 = lr_schedules[0].get_learning_rate(epoch), lr_schedules[1].get_learning_rate(epoch)
        optimizer_all.set_lrs(lr0, lr1)
        order = torch.randperm(n_local)                      # DataLoader(shuffle=True)
        # ONE asynchronous upload of the epoch's order (pinned, double-buffered): a pageable host->device copy per step
        # would make the host wait for the queued GPU work every step
        opin = order_pinned[epoch & 1]
        opin.copy_(order)
        order_dev = order_device[epoch & 1]
        order_dev.copy_(opin, non_blocking=True)

        def draw(it):
            sl = slice(it * local_batch, (it + 1) * local_batch)
            xyz, sdf_gt = cache.sample(order[sl], num_samp_per_scene, generator=gen, scene_ids_device=order_dev[sl])
            return order_dev[sl], xyz, sdf_gt

        nxt = [draw(0)]
        for it in range(steps_per_epoch):
            scenes_dev, xyz, sdf_gt = nxt[0]

            def prefetch(it=it):                             # the next batch's sampling kernel: independent of this step's
                if it + 1 < steps_per_epoch:                 # gradients, so at world > 1 it runs under their all-reduce
                    nxt[0] = draw(it + 1)

            fused(scenes_dev, 2 * int(num_samp_per_scene / 2), xyz, sdf_gt, epoch, lr0, lr1,
                  batch_split=batch_split, n_norm=n_norm, under_allreduce=prefetch, loss_out=loss_buf[it:it + 1])
        lat_mag = None
        if multi:
            torch.distributed.all_reduce(loss_buf)           # per-rank partial losses share the global normaliser
            mag = torch.stack([torch.norm(lat.detach(), dim=1).sum(), torch.full((), float(n_local), device=device)])
            torch.distributed.all_reduce(mag)                # mean code magnitude over the WHOLE table (:548-552), not this shard's
            lat_mag = mag[0] / mag[1]
        stats.pop(loss_log, lat_mag_log, param_mag_log)      # the PREVIOUS epoch's values (its copy finished long ago)
        stats.push(epoch, loss_buf, lat, lat_mag)            # this epoch's: asynchronous, consumed one epoch later
        end = time.time()
        tot_time = time.time() - start_train
        avg = tot_time / (epoch - start_epoch + 1)
        if epoch == num_epochs:
            logging.info(f"Finished {epoch} ({epoch}/{num_epochs}) [{epoch / num_epochs * 100:.2f}%] after "
                         f"{str(datetime.timedelta(seconds=round(tot_time)))}")
        else:
            rem = str(datetime.timedelta(seconds=round(avg * (num_epochs - epoch))))
            logging.info(f"Finished {epoch} ({epoch}/{num_epochs}) [{epoch / num_epochs * 100:.2f}%] in {rem} ({avg:.2f}s/epoch)")
        timing_log.append(end - start)
        lr_log.append([s.get_learning_rate(epoch) for s in lr_schedules])
        if epoch in checkpoints or epoch % log_frequency == 0 or epoch == num_epochs:
            stats.pop(loss_log, lat_mag_log, param_mag_log)  # logs are complete up to this epoch before anything is saved
        if epoch in checkpoints:
            save_all(str(epoch) + ".pth", epoch)
        if epoch % log_frequency == 0:
            save_all("latest.pth", epoch)
            if rank == 0:
                save_logs(experiment_directory, loss_log, lr_log, timing_log, lat_mag_log, param_mag_log, epoch)
    if multi:
        dist.shutdown()
