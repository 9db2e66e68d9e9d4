#!/usr/bin/env python
"""Throughput benchmark of the EfficientSAM3 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)

A "step" = one pass of the hot path over one batch of synthetic input per rank:
  encode  : EV-M (EfficientViT-B1) backbone + student head + BOTH ViTDet FPN necks + conv_s0/s1
            (the full SAM3VLBackbone.forward_image graph; only the x0.5 level that the reference
            computes and immediately discards is not executed) on [32,3,1008,1008] fp32 inputs
            already resident in HBM (BASELINE.json configs[1]: "EV-M bf16 batch=32 @1024^2,
            point+box prompts, 1xMI355X"; the network's native resolution is 1008);
  decode  : prompt encoder + two-way mask decoder for one point+box prompt per image;
  post    : hole filling + bilinear upsample to 1008x1008 + threshold (uint8 masks, on device).
For N > 1 every rank processes its own 32-image shard (weak scaling, no data-path collective in
the model) and rank 0 gathers the uint8 masks over RCCL each step (the only exchange step the
path has).

The JSON line also carries:
  roofline     : the dominant kernel by measured time (since round 3 the level-0 up-conv of the SAM3-side
                 neck: ConvT' o 1x1 o 3x3 composed into one implicit GEMM on gemm256p, 2.78 TFLOP per launch)
                 timed live with HIP events on its launch stream inside the timed steps; `traffic` = HBM bytes
                 of that launch from the committed rocprofv3 PMC passes (profiles/pmc_dominant_kernel.json),
                 `effective_clock_ghz` / `frac_of_peak_at_that_clock` from the SQ / GRBM passes of the same
                 launch (profiles/pmc_dominant_kernel_sq.json): the chip clocks to its power budget under a
                 dense MFMA stream, so the fraction of the 2.4 GHz nominal peak (`frac`) and the fraction of
                 what the matrix pipes can issue at the sustained clock are two different numbers.
  cpu_baseline : the oracle (oracle/ref_model.py, a port of the reference's fp32 CPU path) timed
                 on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 32
FLOPS_PER_IMAGE_G = {  # SURVEY.md §8(d) numerators (GFLOP / image, 2*MAC), EV-M interactive graph
    "backbone": 20.3, "head": 19.9, "necks": 429.8 - 2 * 2.2, "conv_s0s1": 2.04, "decode": 4.49}


def reference_graph_gflop(backbone: str, model: str, text: bool):
    """GFLOP / image of the REFERENCE's layer list for the benched configuration (SURVEY.md §8(d) "Other configs"), or
    None where the survey gives no figure (the S / L student sizes)."""
    if text:  # config 4: ViT-H 5.4 TF + single neck 0.215 + grounding ~0.42 + text 0.0005
        return 6035.5 if (backbone, model) == ("sam3", "vit_h") else None
    table = {("efficientvit", "b1"): sum(FLOPS_PER_IMAGE_G.values()), ("tinyvit", "11m"): 543.2, ("repvit", "m1.1"): 511.1,
             ("sam3", "vit_h"): 5400.0 + 429.8 - 2 * 2.2 + 2.04 + 4.49}
    return table.get((backbone, model))
PEAK_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_HBM_GBS = 8000.0


def cpu_baseline(sample_images: int = 5):
    """Time the oracle on the host cores: set_image + one predict_inst per image."""
    import numpy as np
    import torch

    from efficientsam3_amd import schema, synth
    from oracle import ref_model
    cores = os.cpu_count() or 1
    # 32 threads: the oracle's convolutions / GEMMs at batch 1 are FASTER on 32 threads than on 64 or on torch's default of one per physical
    # core on the 256-CPU GPU hosts (profiles/r06/host_threads_oracle.txt: 14.6 s / 25.5 s / 51.1 s for the same oracle work) -- the baseline
    # is the best the CPU path does here, not a strawman
    threads = min(cores, 32)
    torch.set_num_threads(threads)
    sd = schema.synthetic_state_dict("efficientvit", "b1", seed=0)
    pts, labels, boxes = synth.prompts(sample_images + 1, seed=2)
    times = []
    for i in range(sample_images + 1):  # first one is warm-up
        img = synth.smooth_image_u8(seed=100 + i)
        x = torch.from_numpy(synth.normalise_to_chw_f32(img))[None]
        t0 = time.perf_counter()
        with torch.inference_mode():
            st = ref_model.set_image(sd, x, (1008, 1008), "b1")
            ref_model.predict_inst(sd, st, point_coords=pts[i], point_labels=labels[i], box=boxes[i],
                                   multimask_output=False)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    return {"value": round(1.0 / t, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "seconds_per_image": [round(x, 3) for x in times[1:]],
            "sample": f"median of {sample_images} images (after 1 warm-up) of the same synthetic workload, "
                      f"fp32 oracle, set_image + predict_inst(point+box), torch CPU {threads} threads"}


def parse_cpulist(spec: str) -> set:
    """sysfs cpulist syntax ("0-63,128-191", "5", "") -> set of cpu numbers"""
    cpus = set()
    for part in spec.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        elif part:
            cpus.add(int(part))
    return cpus


def bind_to_gpu_numa_node(local_rank: int):
    """Pin this rank's host threads to the CPUs next to its GPU (sysfs: /sys/bus/pci/devices/<bdf>/local_cpulist): the per-rank
    staging copies and worker threads of an N-rank job then stay on the socket the GPU hangs off (host IO, not
    xGMI, is the scaling risk).  Returns a short description for the JSON line, or None where the topology is not exposed."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            spec = f.read().strip()
        cpus = parse_cpulist(spec) & os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return f"{bdf}: {len(cpus)} cpus ({spec})"
    except Exception:  # noqa: BLE001  (no sysfs entry in a container, attribute missing): run unbound
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="images per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the legs reported under config (two batches in flight, PCIe-inclusive, API level): profiler passes want one "
                         "stream and one dominant launch per step")
    ap.add_argument("--workload", default="interactive", choices=["interactive", "text"],
                    help="interactive: set_image_batch + predict_inst (headline, BASELINE configs[1]); text: image encoder + "
                         "MobileCLIP-S0 text encoder + PCS grounding detector, one text prompt per image (configs[3] with "
                         "--backbone sam3 --model vit_h --batch 8)")
    ap.add_argument("--backbone", default="efficientvit", help="student family (the headline metric is efficientvit/b1)")
    ap.add_argument("--model", default="b1")
    ap.add_argument("--no-fuse", action="store_true",
                    help="run the reference's layer list without composing ConvT->1x1 / 3x3->conv_s0,s1")
    ap.add_argument("--dry-collective", action="store_true",
                    help="single GPU only: create a ONE-rank RCCL process group and drive the per-step mask gather through it "
                         "(private buffers, side stream, dist.gather) exactly as an N-rank job does; the line then reports "
                         "collective_backend nccl / ranks_in_process_group 1.  A plumbing check, not a scaling number")
    ap.add_argument("--sam2-only", action="store_true",
                    help="consumer-minimal graph (skip the sam3 neck); NOT the headline number")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from efficientsam3_amd import build_efficientsam3_image_model, build_sam3_image_model, schema, synth
    from efficientsam3_amd import dist as esdist

    rank, local_rank, world = esdist.env_ranks()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched from a bare shell (`python bench.py --gpus N`): re-exec as N ranks, one per GPU, over RCCL
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execvpe(cmd[0], cmd, env)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    affinity = bind_to_gpu_numa_node(local_rank) if world > 1 else None
    dry_coll = args.dry_collective and world == 1
    esdist.init_process_group("nccl", dev, force=dry_coll)

    text = args.workload == "text"
    sd = schema.synthetic_state_dict(args.backbone, args.model, seed=0, enable_inst_interactivity=not text)
    tkw = {}
    if text:
        sd.update(schema.synthetic_text_state_dict("MobileCLIP-S0", 16, seed=0))
        sd.update(schema.synthetic_pcs_state_dict(seed=0))
        tkw = dict(text_encoder_type="MobileCLIP-S0", text_encoder_context_length=16)
    if args.backbone == "sam3":  # ViT-H teacher (not the headline configuration)
        model = build_sam3_image_model(device=dev, enable_inst_interactivity=not text, dtype=args.dtype, state_dict=sd,
                                       dual_neck=not args.sam2_only, fuse_linear_chains=not args.no_fuse, **tkw)
    else:
        model = build_efficientsam3_image_model(device=dev, enable_inst_interactivity=not text,
                                                backbone_type=args.backbone, model_name=args.model,
                                                dtype=args.dtype, state_dict=sd, dual_neck=not args.sam2_only,
                                                fuse_linear_chains=not args.no_fuse, **tkw)
    eng = model.engine
    B = args.batch
    # synthetic batch, resident in HBM before the timed region: 4 distinct images tiled to B
    base = [synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=1 + 7 * rank)),
            synth.normalise_to_chw_f32(synth.noise_image_u8(seed=2 + 7 * rank)),
            synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=3 + 7 * rank)),
            synth.normalise_to_chw_f32(synth.noise_image_u8(seed=4 + 7 * rank))]
    x = torch.from_numpy(np.stack([base[i % 4] for i in range(B)])).to(dev)
    pts, labels, boxes = synth.prompts(B, seed=2 + rank)
    coords, labs = model._prep_prompts(pts, labels, boxes, True, (1008, 1008))  # [B,3,2] box+point
    c_d = torch.from_numpy(coords).to(dev)
    l_d = torch.from_numpy(labs).to(dev)
    pi_d = torch.arange(B, dtype=torch.int32, device=dev)
    bufs = {"enc": None, "dec": None, "post": None}
    # text workload: one 16-token prompt per image (seeded synthetic token ids: <sot>, 1-4 word ids, <eot>, padding)
    rng = np.random.default_rng(5 + rank)
    tok = np.zeros((B, 16), dtype=np.int64)
    for i in range(B):
        n = int(rng.integers(1, 5))
        tok[i, 0], tok[i, 1:1 + n], tok[i, 1 + n] = 49406, rng.integers(300, 40000, size=n), 49407
    tok_d = torch.from_numpy(tok).to(dev)

    def step_text():
        bufs["enc"] = out = eng.encode(x, want_sam3=True, want_sam2=False, out=bufs["enc"])
        mem_t, _ = eng.encode_text(tok_d)
        g = eng.ground(out["sam3_fpn"], mem_t, tok_d == 0)
        return g["pred_masks"], g["pred_logits"]

    def step():
        if text:
            return step_text()
        # fixed output buffers, nothing is allocated inside the timed region
        bufs["enc"] = out = eng.encode(x, want_sam3=not args.sam2_only, want_sam2=True, out=bufs["enc"])
        bufs["dec"] = low, iou = eng.decode(out["sam2_fpn"], pi_d, c_d, l_d, multimask_output=False, out=bufs["dec"])
        bufs["post"] = masks = eng.postprocess(low, (1008, 1008), return_logits=False, out=bufs["post"])
        if gatherer is not None and coll_err[0] is None:
            # the path's only exchange step: uint8 masks of every shard -> rank 0, on a side stream.  A failing collective
            # must not cost the whole run its number: the error is recorded in the line and the steps go on without it.
            try:
                gatherer.submit(masks)
            except Exception as e:  # noqa: BLE001
                coll_err[0] = f"{type(e).__name__}: {e}"[:300]
        return masks, iou

    coll_err = [None]

    gatherer = esdist.MaskGatherer(dst=0, force_collective=dry_coll) if (world > 1 or dry_coll) and not text else None

    def sync():
        if gatherer is not None and coll_err[0] is None:
            try:
                gatherer.flush()  # the last step's gather is part of the timed work
            except Exception as e:  # noqa: BLE001
                coll_err[0] = f"{type(e).__name__}: {e}"[:300]
        if world > 1 or dry_coll:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- setup (not warm-up): one step with EVERY launch bracketed by HIP events gives the per-stage
    # split and names the dominant launch; in the timed region only that launch carries events, so the
    # instrumentation does not slow the step down.
    step()          # cold start (workspace sizing, first touch of every buffer) stays out of the per-stage table
    sync()
    eng.profile_enable(True)
    step()
    sync()
    prof_all = eng.profile_report()
    eng.profile_enable(False)
    dom_tag = prof_all[0]["tag"] if prof_all else None
    eng.profile_tag(dom_tag)

    for _ in range(args.warmup):
        step()
    sync()
    eng.profile_tag(dom_tag)  # clears the records: only launches of the timed steps remain
    t0 = time.perf_counter()
    for _ in range(args.steps):
        masks, iou = step()
    sync()
    elapsed = time.perf_counter() - t0
    prof_dom = eng.profile_report()
    eng.profile_tag(None)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fg = float((masks > 0).float().mean().item()) if text else float(masks.float().mean().item())

    # ---- two batches in flight (reported in config, never `value`): the same step from TWO engine replicas on two HIP streams, one Python
    # thread each.  `value` above is one stream running its steps back to back (every round's definition); a serving process that keeps two
    # batches in flight lets one batch's latency-bound backbone / decoder kernels fill the tails and stalls of the other's, and loses
    # nothing on the power-bound GEMM launches (profiles/r06/two_stream_probe.txt, gemm_grid_cap.txt).  Same work per step, same results.
    model2 = two_ips = None
    if rank == 0 and world == 1 and not text and args.backbone != "sam3" and not args.headline_only:
        import threading
        model2 = build_efficientsam3_image_model(device=dev, enable_inst_interactivity=True, backbone_type=args.backbone,
                                                 model_name=args.model, dtype=args.dtype, state_dict=sd,
                                                 dual_neck=not args.sam2_only, fuse_linear_chains=not args.no_fuse)

        def make_step(mdl):
            e_, b_ = mdl.engine, {"enc": None, "dec": None, "post": None}
            x_ = x.clone()

            def one():
                b_["enc"] = o_ = e_.encode(x_, want_sam3=not args.sam2_only, want_sam2=True, out=b_["enc"])
                b_["dec"] = lw, _ = e_.decode(o_["sam2_fpn"], pi_d, c_d, l_d, multimask_output=False, out=b_["dec"])
                b_["post"] = e_.postprocess(lw, (1008, 1008), return_logits=False, out=b_["post"])
                return b_["post"]
            return one

        steps2 = [make_step(model), make_step(model2)]
        per = max(2, args.steps // 2)
        gate2 = threading.Barrier(3)
        means, errs2 = [None, None], []

        def flight(i):
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    for _ in range(2):
                        steps2[i]()
                    torch.cuda.current_stream(dev).synchronize()
                    gate2.wait()
                    o_ = None
                    for _ in range(per):
                        o_ = steps2[i]()
                    torch.cuda.current_stream(dev).synchronize()
                    gate2.wait()
                    means[i] = float(o_.float().mean().item())
            except Exception as e:  # noqa: BLE001
                errs2.append(f"{type(e).__name__}: {e}"[:200])
                gate2.abort()

        th2 = [threading.Thread(target=flight, args=(i,), daemon=True) for i in range(2)]
        for t_ in th2:
            t_.start()
        try:
            gate2.wait()
            t4 = time.perf_counter()
            gate2.wait()
            two_ips = 2 * per * B / (time.perf_counter() - t4)
        except threading.BrokenBarrierError:
            two_ips = None
        for t_ in th2:
            t_.join(timeout=60)
        if two_ips is not None and (errs2 or means[0] != means[1] or abs(means[0] - fg) > 1e-6):
            two_ips = None    # both replicas must reproduce the timed region's masks

    # ---- PCIe-inclusive leg (reported in config, never `value`): B uint8 1024x1024 HWC images in pinned host
    # memory -> one H2D copy -> device antialiased resize to 1008^2 + normalise (P1) -> the same step
    host_incl = None
    if not text and not args.headline_only:   # every rank runs it (N ranks pull their shards over their own PCIe links at the same time)
        u8 = torch.from_numpy(np.random.default_rng(rank).integers(0, 256, (B, 1024, 1024, 3), dtype=np.uint8)).pin_memory()
        # two device buffers: batch k + 1 crosses the link on a side stream while batch k is resized, encoded and decoded (a
        # streaming caller's pipeline; round 5 issued the copy on the compute stream, so the device idled for every transfer)
        u8_d = [torch.empty_like(u8, device=dev) for _ in range(2)]
        x_keep = x
        cur = torch.cuda.current_stream(dev)
        h2d = torch.cuda.Stream(device=dev)
        arrived, consumed = [None, None], [None, None]

        def send(k):      # host -> device copy of batch k into buffer k % 2, once the resize that last read that buffer has run
            with torch.cuda.stream(h2d):
                if consumed[k % 2] is not None:
                    h2d.wait_event(consumed[k % 2])
                u8_d[k % 2].copy_(u8, non_blocking=True)
                arrived[k % 2] = torch.cuda.Event()
                arrived[k % 2].record(h2d)

        def step_from_host(k):
            nonlocal x
            cur.wait_event(arrived[k % 2])
            eng.preprocess_resize_u8_batch(u8_d[k % 2], x)   # one launch for the equal-sized batch
            consumed[k % 2] = torch.cuda.Event()
            consumed[k % 2].record(cur)
            send(k + 1)
            return step()

        send(0)
        step_from_host(0)
        sync()
        reps = 6
        t1 = time.perf_counter()
        send(1)           # the timed region pays its first transfer in full: nothing is in flight when it starts
        for k in range(1, reps + 1):
            step_from_host(k)
        sync()
        el1 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([el1], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el1 = float(t.item())
        host_incl = world * B * reps / el1
        x = x_keep

    # ---- API-level leg (reported in config, never `value`): the reference-shaped Python calls a user makes --
    # Sam3Processor.set_image_batch(list of PIL images) + model.predict_inst_batch(point + box per image) -> numpy masks at
    # the ORIGINAL 1024x1024 size, i.e. including PIL -> tensor conversion, H2D, device resize, and the D2H copies of
    # float32 masks / IoU scores / low-res logits that the reference's contract returns
    api_ips = api2_ips = api2_err = None
    if rank == 0 and world == 1 and not text and model2 is not None:
        try:
            from PIL import Image
            from efficientsam3_amd import Sam3Processor
            rng_img = np.random.default_rng(0).integers(0, 256, (4, 1024, 1024, 3), dtype=np.uint8)
            pil = [Image.fromarray(rng_img[i % 4]) for i in range(B)]
            proc = Sam3Processor(model)
            sx = 1024.0 / 1008.0
            pcs = [pts[i] * sx for i in range(B)]
            bxs = [boxes[i] * sx for i in range(B)]

            def api_step():
                st = proc.set_image_batch(pil)
                return model.predict_inst_batch(st, point_coords_batch=pcs, point_labels_batch=[labels[i] for i in range(B)],
                                                box_batch=bxs, multimask_output=False)

            for _ in range(3):   # pinned staging buffers, result pool and the worker threads reach their steady state
                api_step()
            sync()
            reps = 5
            t2 = time.perf_counter()
            for _ in range(reps):
                m_api, _, _ = api_step()
            sync()
            api_ips = B * reps / (time.perf_counter() - t2)
            assert len(m_api) == B and m_api[0].shape == (1, 1024, 1024)
            # The same calls from TWO caller threads, each with its own model replica (engine handle, workspace, stream) on this GPU:
            # the API is synchronous (every predict_inst_batch ends with its masks on the host), so ONE caller cannot overlap batch k's
            # hand-back with batch k + 1's encode -- two callers do, with the reference's API unchanged.  Reported beside the
            # one-caller figure, never instead of it.
            import threading
            callers = [(model, proc), (model2, Sam3Processor(model2))]
            lbl = [labels[i] for i in range(B)]
            reps2, errs = 6, []
            gate = threading.Barrier(len(callers) + 1)

            def caller(mdl, prc):
                try:
                    torch.cuda.set_device(dev)
                    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                        def one():
                            st = prc.set_image_batch(pil)
                            return mdl.predict_inst_batch(st, point_coords_batch=pcs, point_labels_batch=lbl, box_batch=bxs,
                                                          multimask_output=False)
                        for _ in range(3):
                            one()
                        torch.cuda.current_stream(dev).synchronize()
                        gate.wait()
                        out = None
                        for _ in range(reps2):
                            out = one()
                        torch.cuda.current_stream(dev).synchronize()
                        gate.wait()
                        assert len(out[0]) == B
                except Exception as e:  # noqa: BLE001
                    errs.append(f"{type(e).__name__}: {e}"[:200])
                    gate.abort()

            ths = [threading.Thread(target=caller, args=c, daemon=True) for c in callers]
            for t_ in ths:
                t_.start()
            try:
                gate.wait()
                t3 = time.perf_counter()
                gate.wait()
                api2_ips = len(callers) * B * reps2 / (time.perf_counter() - t3)
            except threading.BrokenBarrierError:
                api2_ips = None
            for t_ in ths:
                t_.join(timeout=60)
            api2_err = errs[0] if errs else None
        except ImportError:
            api_ips = None

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        # ---- roofline of the dominant kernel (by measured time) --------------------------------
        prof = prof_all                       # per-launch table of the one eager profiling step (setup)
        dom = prof_dom[0] if prof_dom else None  # the dominant launch, timed live in every timed step
        roof = None
        if dom is not None:
            avg_ms = dom["ms"] / dom["launches"]
            tflops = dom["algorithmic_flops"] / (avg_ms * 1e-3) / 1e12
            gbs = dom["algorithmic_bytes"] / (avg_ms * 1e-3) / 1e9
            mfma_bound = dom["algorithmic_flops"] / (PEAK_BF16_TFLOPS * 1e12) >= dom["algorithmic_bytes"] / (PEAK_HBM_GBS * 1e9)
            traffic = None
            traffic_source = None   # `traffic` is NOT measured in this run: PMC counters need rocprofv3 around the process
            pmc_path = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")
            if os.path.exists(pmc_path):
                try:  # the committed PMC pass covers the headline config's dominant launch (its tag is recorded in the file)
                    pmc = json.load(open(pmc_path))
                    if dom["tag"] == pmc.get("tag") and B == BATCH and not text:
                        traffic = pmc.get("hbm_bytes_per_launch")
                        traffic_source = (f"committed rocprofv3 --pmc pass profiles/pmc_dominant_kernel.json (round {pmc.get('round')}, "
                                          f"{pmc.get('command', 'tools/gpu_profile_round.sh')}): FETCH_SIZE x 2 + WRITE_SIZE per launch of this "
                                          "kernel and shape, not a counter read in this run")
                except Exception:
                    traffic = None
            if mfma_bound:
                roof = {"bound": "mfma", "achieved": round(tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tflops / PEAK_BF16_TFLOPS, 4), "traffic": traffic}
            else:
                roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": traffic}
            sq_path = os.path.join(ROOT, "profiles", "pmc_dominant_kernel_sq.json")
            if mfma_bound and traffic is not None and os.path.exists(sq_path):
                try:
                    dvd = json.load(open(sq_path)).get("derived", {})
                    if dvd.get("dense_bf16_peak_at_that_clock_tflops"):
                        roof["effective_clock_ghz"] = dvd.get("effective_clock_ghz")
                        roof["mfma_pipe_busy_fraction"] = dvd.get("mfma_pipe_busy_fraction")
                        roof["frac_of_peak_at_that_clock"] = round(tflops / dvd["dense_bf16_peak_at_that_clock_tflops"], 4)
                except Exception:
                    pass
            roof["traffic_source"] = traffic_source
            roof.update(kernel=dom.get("kernel", dom["tag"]), tag=dom["tag"], launches_per_step=dom["launches"] // args.steps,
                        timed_launches=dom["launches"],
                        avg_launch_ms=round(avg_ms, 4), algorithmic_flops_per_launch=dom["algorithmic_flops"],
                        algorithmic_bytes_per_launch=dom["algorithmic_bytes"])
        gf_ref = reference_graph_gflop(args.backbone, args.model, text)  # reference layer list
        # executed GFLOP/image, measured from the per-launch algorithmic flop counts of this run
        gf_img = sum(p_.get("algorithmic_flops_total", p_["algorithmic_flops"] * p_["launches"]) for p_ in prof) / B / 1e9
        total_k = sum(p["ms"] for p in prof)
        # the whole step against its own floors: every launch priced at max(flops / MFMA peak, bytes / HBM peak) of ITS algorithmic work
        floor_ms = sum(max(p_.get("algorithmic_flops_total", p_["algorithmic_flops"] * p_["launches"]) / (PEAK_BF16_TFLOPS * 1e12),
                           p_.get("algorithmic_bytes_total", p_["algorithmic_bytes"] * p_["launches"]) / (PEAK_HBM_GBS * 1e9)) * 1e3 for p_ in prof)
        stage_ms = {}
        for p_ in prof:
            t_ = p_["tag"]
            key = ("neck" if ".convs." in t_ or ".sam2_convs." in t_ or "conv_s0" in t_ or "conv_s1" in t_ or t_ == "resize_shuffle"
                   else "head" if ".head." in t_
                   else "grounding" if t_.startswith(("pcs_", "transformer.", "geometry_encoder.", "segmentation_head.", "dot_prod_scoring."))
                   else "text" if "language_backbone" in t_ or t_.startswith(("text_", "seq_dwconv", "bsc_to_sbc"))
                   else "backbone" if "trunk.model.backbone" in t_ or "vision_backbone.trunk." in t_ and ".head." not in t_
                   or t_.startswith(("dwconv", "stem", "lite_mla", "grouped_pw", "resize", "mbconv_fused", "squeeze_excite",
                                     "window_attn", "vit_", "patchify"))
                   else "decode+post")
            stage_ms[key] = stage_ms.get(key, 0.0) + p_["ms"]
        if world > 1:   # an N-rank line without a live process group behind it is not a scaling number
            assert dist.is_initialized() and dist.get_world_size() == world and dist.get_backend() == "nccl", \
                "multi-GPU run without an RCCL process group of WORLD_SIZE ranks"
        out = {
            "metric": "images/sec encode+decode @1024^2 (EV-M bf16)" if (args.backbone, args.model, text) == ("efficientvit", "b1", False)
            else (f"images/sec text-prompted encode+ground @1024^2 ({args.backbone}-{args.model} + MobileCLIP-S0-16 {args.dtype})" if text
                  else f"images/sec encode+decode @1024^2 ({args.backbone}-{args.model} {args.dtype})"), "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic (seeded images at the network's native 1008x1008, seeded realistic random-init weights)",
            "config": {"workload": ("EV-M (EfficientViT-B1)" if args.backbone == "efficientvit" else f"{args.backbone}-{args.model}") + (" image encoder + text encoder + PCS grounding detector (200 queries -> logits, boxes, "
                                                       "288x288 mask logits), one text prompt per image, " if text else
                                                       " engine-level encode + decode(point+box) + postprocess per image (the kernels behind "
                                                       "set_image_batch + predict_inst; the Python API leg is config.api_level_*), ")
                                   + f"batch={B} per GPU, " + ("sam3 neck" if text else "full dual-neck graph") + (" [sam2-only variant]" if args.sam2_only else ""),
                       "global_batch": world * B, "resolution": 1008, "prompts_per_image": 1,
                       "parallelism": f"dp{world} (image shards, RCCL gather of uint8 masks on a side stream)" if world > 1 else
                       ("single GPU, one-rank RCCL group: the per-step mask gather runs through dist.gather on a side stream (--dry-collective)"
                        if dry_coll else "single GPU"),
                       "side_stream_gathers": None if gatherer is None else gatherer.side_stream_gathers,
                       "collective_error": coll_err[0],
                       "ranks_in_process_group": (dist.get_world_size() if dist.is_initialized() else 1),
                       "collective_backend": (dist.get_backend() if dist.is_initialized() else None),
                       "cpu_affinity_rank0": affinity,
                       "graph": ("reference layer list" if args.no_fuse else
                                 "linear chains composed at load time (ConvT∘1x1, 3x3∘conv_s0/s1): same outputs, fewer FLOPs"),
                       "gflop_per_image_executed": round(gf_img, 1), "gflop_per_image_reference_graph": None if gf_ref is None else round(gf_ref, 1),
                       "end_to_end_mfma_frac": round(value * gf_img * 1e9 / (world * PEAK_BF16_TFLOPS * 1e12), 4),
                       "kernel_ms_per_step_by_stage": {k: round(v, 3) for k, v in sorted(stage_ms.items())},
                       "kernel_ms_per_step_total": round(total_k, 3), "kernel_floor_ms_per_step": round(floor_ms, 3),
                       "launches_per_step": sum(p_["launches"] for p_ in prof),
                       "kernel_ms_note": "per-stage kernel times come from one fully event-instrumented step before the "
                                         "timed region; in the timed steps only the dominant launch carries HIP events",
                       "pcie_inclusive_images_per_s": None if host_incl is None else round(host_incl, 1),
                       "pcie_inclusive_note": "double-buffered: batch k + 1 travels on a side stream under batch k's step; uint8 1024x1024 HWC batch in pinned host memory -> H2D -> device resize to 1008^2 "
                                              "+ normalise -> the same step, on every rank at once (aggregate over the ranks, "
                                              "slowest rank's clock); measured after the timed region, not `value`",
                       "api_level_images_per_s": None if api_ips is None else round(api_ips, 1),
                       "two_batches_in_flight_images_per_s": None if two_ips is None else round(two_ips, 1),
                       "two_batches_in_flight_note": "the same step from two engine replicas on two HIP streams (one Python thread each), same "
                                                     "results; `value` is ONE stream running its steps back to back",
                       "api_level_two_callers_images_per_s": None if api2_ips is None else round(api2_ips, 1),
                       "api_level_two_callers_note": "the same calls from two Python threads, one model replica each on this GPU (the API is "
                                                     "synchronous: one caller cannot overlap a batch's hand-back with the next batch's encode)"
                                                     + ("" if api2_err is None else f"; FAILED: {api2_err}"),
                       "api_level_note": "Sam3Processor.set_image_batch(32 PIL 1024x1024 images) + model.predict_inst_batch(point+box) -> "
                                         "numpy float32 masks at 1024x1024, IoU scores, low-res logits (the reference's return contract, D2H "
                                         "included); measured after the timed region, not `value`",
                       "mask_fg_fraction": round(fg, 4), "workspace_gb": round(eng.workspace_bytes() / 2 ** 30, 2)},
            "roofline": roof,
            "step_roofline_frac": round(floor_ms / total_k, 4) if total_k > 0 else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if os.environ.get("ESAM3_BENCH_PROFILE_OUT"):
            with open(os.environ["ESAM3_BENCH_PROFILE_OUT"], "w") as f:
                json.dump({"per_tag": prof, "steps": 1, "batch": B, "dominant_timed": prof_dom}, f, indent=1)
    if dist.is_initialized():
        if gatherer is not None and rank == 0 and coll_err[0] is None:
            got = gatherer.result()
            assert got is not None and got.shape[0] == world * B, "mask gather did not deliver every shard"
        dist.destroy_process_group()
    if rank == 0:
        # The JSON line is the LAST thing this process writes: the process group is gone, and whatever RCCL left in the C
        # stdio buffers (its version banner goes to stdout and would otherwise be flushed after this line at exit) is out.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
