"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/effconf.h declares,
the host mirror keeps the reference's state-dict surface and config rules, and the product path refuses
to run without a GPU tensor (no silent fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from efficientconformer_amd import ConformerEncoder, ModelCTC, Transducer, _lib, named_config, params, synth
from efficientconformer_amd.config import build_plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    hdr = open(os.path.join(ROOT, "include", "effconf.h")).read()
    declared = sorted(set(re.findall(r"\b(effconf_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libeffconf.so does not export %s" % name
    assert sorted(_lib.SIGNATURES) == declared, "ctypes binding out of sync with include/effconf.h"
    assert _lib.load().effconf_abi_version() == _lib.ABI_VERSION == 3
    # the diagnostics live in a SEPARATE library (include/effconf_debug.h, libeffconf_debug.so = the product objects + csrc/debug.hip + the packed-fp32 mel
    # build): the product library exports none of them, the diagnostic one exports them and the whole product ABI
    dhdr = open(os.path.join(ROOT, "include", "effconf_debug.h")).read()
    ddecl = sorted(set(re.findall(r"\b(effconf_debug_[a-z_0-9]+)\s*\(", dhdr)))
    assert len(ddecl) == 10 and sorted(_lib.DEBUG_SIGNATURES) == ddecl
    assert not [n for n in declared if n.startswith("effconf_debug")]
    exported = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "effconf_debug" not in exported and "debug_neighbour" not in exported
    # -fvisibility=hidden + the headers' visibility pragma: the dynamic symbol table holds the declared C entries and nothing else
    # (before round 6: 58 mangled internals - launch_chain, ec_fail, std::map instantiations - were linkable beside them)
    syms = [ln.split()[-1] for ln in exported.splitlines() if ln.strip() and ln.split()[-2] in "TtWwVvBbDdRr"]
    assert syms and sorted(set(syms)) == declared, [n for n in syms if n not in declared][:10]
    dexp = subprocess.run(["nm", "-D", "--defined-only", os.path.join(os.path.dirname(_lib.LIB_PATH), "libeffconf_debug.so")], capture_output=True, text=True).stdout
    dsyms = sorted(set(ln.split()[-1] for ln in dexp.splitlines() if ln.strip()))
    assert dsyms == sorted(declared + ddecl), [n for n in dsyms if n not in declared + ddecl][:10]
    dlib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libeffconf_debug.so"))
    for name in declared + ddecl:
        assert hasattr(dlib, name), "libeffconf_debug.so does not export %s" % name


@pytest.mark.parametrize("ksize,channels", [(15, 120), (31, 176), (7, 10)])
def test_depthwise_toeplitz_table_of_the_matrix_pipe_kernel(ksize, channels):
    """pack_dwconv_mfma (csrc/conv.hip, host code; through the diagnostic library's effconf_debug_pack_dwconv_mfma): lane 4 b + i of dwconv_mfma_kernel holds, per
    group q of 4 taps, row i of the Toeplitz block A_q[i][k] = w[4 q + k - i] as bf16 hi | lo halves.  Checked against numpy: hi + lo reproduce every folded fp32
    tap to 2^-16 of its magnitude, entries outside the taps are zero, and - the property the kernel relies on - the blocks rebuild the convolution:
    out[4 j + i] = sum_q sum_k A_q[i][k] x[4 (j + q) + k] equals sum_m w[m] x[4 j + i + m] for random x."""
    import ctypes
    dlib = _lib.load_debug()
    rng = np.random.default_rng(ksize)
    w = (rng.standard_normal((ksize, channels)) * np.exp(rng.uniform(-6, 2, size=(1, channels)))).astype(np.float32)
    nq = (ksize + 6) // 4
    dst = np.zeros((channels, 4, nq, 2, 4), dtype=np.uint16)
    _lib.check(dlib.effconf_debug_pack_dwconv_mfma(w.ctypes.data_as(ctypes.c_void_p), ksize, channels, dst.ctypes.data_as(ctypes.c_void_p), dst.size), "pack", dlib)
    assert dlib.effconf_debug_pack_dwconv_mfma(w.ctypes.data_as(ctypes.c_void_p), 9, channels, dst.ctypes.data_as(ctypes.c_void_p), dst.size) != 0      # unsupported size
    val = (dst.astype(np.uint32) << 16).view(np.float32)                     # [ch][i][q][hi / lo][k]
    a = val[:, :, :, 0, :].astype(np.float64) + val[:, :, :, 1, :].astype(np.float64)
    for i in range(4):
        for q in range(nq):
            for k in range(4):
                tap = 4 * q + k - i
                if 0 <= tap < ksize:
                    assert np.all(np.abs(a[:, i, q, k] - w[tap]) <= np.abs(w[tap]) * 2.0 ** -16 + 1e-38), (i, q, k)
                else:
                    assert not dst[:, i, q, :, k].any()
    x = rng.standard_normal(4 * (3 + nq)).astype(np.float64)
    for ch in (0, channels // 2, channels - 1):
        for j in range(3):
            for i in range(4):
                blocks = sum(a[ch, i, q, k] * x[4 * (j + q) + k] for q in range(nq) for k in range(4))
                direct = sum(float(w[m, ch]) * x[4 * j + i + m] for m in range(ksize))
                assert abs(blocks - direct) <= 1e-4 * max(1.0, abs(direct)) * max(1.0, float(np.abs(w[:, ch]).max()))


def test_create_rejects_bad_config_and_reports_error():
    lib = _lib.load()
    cfg = _lib.EcConfig()
    assert not lib.effconf_encoder_create(ctypes.byref(cfg))
    assert b"config" in lib.effconf_last_error()


@pytest.mark.parametrize("name", ["Tiny", "EfficientConformerCTCSmall", "EfficientConformerCTCMedium",
                                  "EfficientConformerCTCLarge", "EfficientConformerTransducerMedium"])
def test_state_dict_surface_matches_reference_keys(name):
    cfg = named_config(name)
    m = ModelCTC.from_config(cfg)
    plan = m.encoder.plan
    specs = params.param_specs(plan)
    sd = m.encoder.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in specs]
    for k, shape, _ in specs:
        assert tuple(sd[k].shape) == tuple(shape), k
    full = synth.make_state_dict(plan, 0, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    # DDP-saved checkpoints carry ".module." (reference model.py:367-370); torchaudio buffers are ignored
    ddp = {k.replace("encoder.", "encoder.module.", 1): torch.from_numpy(v) for k, v in full.items()}
    ddp["encoder.preprocessing.Spectrogram.window"] = torch.zeros(400)
    m.load_state_dict(ddp)
    assert torch.equal(m.encoder.linear.weight, torch.from_numpy(full["encoder.linear.weight"]))


def test_param_counts_match_reference_readme():
    # README.md:92-94 / SURVEY.md section 6 (encoder + fc, BatchNorm running stats excluded)
    for name, want in (("EfficientConformerCTCSmall", 13281856), ("EfficientConformerCTCMedium", 31570932),
                       ("EfficientConformerCTCLarge", 125675272)):
        m = ModelCTC.from_config(named_config(name))
        assert sum(p.numel() for p in m.parameters()) == want


def test_block_plan_rule():
    plan = build_plan(named_config("EfficientConformerCTCSmall")["encoder_params"])
    b4, b5, b9, b10 = plan.blocks[4], plan.blocks[5], plan.blocks[9], plan.blocks[10]
    assert (b4.dim_model, b4.dim_expand, b4.group_size, b4.conv_stride, b4.max_pos) == (120, 168, 3, 2, 10000)
    assert (b5.dim_model, b5.dim_expand, b5.group_size, b5.conv_stride, b5.max_pos) == (168, 168, 1, 1, 5000)
    assert (b9.dim_model, b9.dim_expand, b9.conv_stride) == (168, 240, 2) and b10.max_pos == 2500
    assert plan.blocks[0].dim_head == 90 and b5.dim_head == 42 and b10.dim_head == 60
    assert plan.lengths(160000) == (1001, 501, [501] * 4 + [251] * 5 + [126] * 6)


def test_unsupported_configs_raise():
    p = named_config("EfficientConformerCTCSmall")["encoder_params"]
    for k, v in (("subsampling_module", "VGG"), ("relative_pos_enc", False), ("linear_att", True)):
        q = dict(p); q[k] = v
        with pytest.raises(NotImplementedError):
            build_plan(q)
    # streaming / causal contexts are native since round 3 (encoders.py:68, 94): the plan carries them, block b's mask stride included
    q = dict(p, causal=True, left_context=64)
    plan = build_plan(q)
    assert plan.causal and plan.left_context == 64 and plan.right_context == 0
    assert [b.mask_stride for b in plan.blocks] == [1] * 5 + [2] * 5 + [4] * 5
    assert build_plan(p).right_context == p["max_pos_encoding"] and not build_plan(p).causal
    q = dict(p); q["subsampling_module"] = "Nope"
    with pytest.raises(Exception):
        build_plan(q)


def test_no_cpu_fallback():
    m = ConformerEncoder(named_config("Tiny")["encoder_params"])
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(2, 16000), torch.tensor([16000, 8000]))
    m.train()
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 16000))


def test_synth_is_deterministic_and_libri_shaped():
    a = synth.make_tensor("linear.weight", (4, 8), "weight", 0)
    b = synth.make_tensor("linear.weight", (4, 8), "weight", 0)
    assert np.array_equal(a, b) and abs(float(a.std()) - 8 ** -0.5) < 0.2
    lens = synth.libri_lengths(256)
    assert lens[0] >= lens[-1] and lens.min() >= 24000 and lens.max() <= 256000
    x = synth.make_audio(lens[:4])
    assert x.shape == (4, lens[0]) and np.all(x[3, lens[3]:] == 0) and np.abs(x).max() <= 1.0


def test_transducer_state_dict_surface_and_config_checks():
    """decoder.* / joint_network.* keys as the reference registers them (decoders.py:46-47, joint_networks.py:41-52)."""
    cfg = named_config("EfficientConformerTransducerMedium")
    m = Transducer.from_config(cfg)
    specs = params.transducer_specs(m.encoder.plan.dim_out, cfg["decoder_params"], cfg["joint_params"])
    sd = m.state_dict()
    for k, shape, _ in specs:
        assert tuple(sd[k].shape) == tuple(shape), k
    assert [k for k in sd if not k.startswith("encoder.")] == [k for k, _, _ in specs]
    syn = synth.make_transducer_state_dict(m.encoder.plan.dim_out, cfg["decoder_params"], cfg["joint_params"], 0, blank_bias=1.2)
    assert np.all(syn["decoder.embedding.weight"][0] == 0)
    with pytest.raises(NotImplementedError):
        Transducer.from_config(dict(cfg, joint_params={"joint_mode": "concat", "dim_model": 640, "act": "tanh"}))
    with pytest.raises(NotImplementedError):
        Transducer.from_config(dict(cfg, decoder_params=dict(cfg["decoder_params"], arch="Transformer")))
    with pytest.raises(NotImplementedError):
        m.forward(None)
    with pytest.raises(RuntimeError):
        m.decode_encoded(torch.zeros(1, 4, 360), None)          # CPU tensor: no fallback
    lib = _lib.load()
    bad = _lib.EcRnntConfig(360, 640, 640, 1000, 2, 5, 0, 0)
    assert not lib.effconf_rnnt_create(ctypes.byref(bad)) and b"num_layers" in lib.effconf_last_error()


def test_checkpoint_roundtrip_reference_layout(tmp_path):
    """Reference checkpoint dict layout (model.py:345-384): DDP '.module.' infix stripped when is_distributed, tokenizer restored,
    optimizer / step ignored."""
    from efficientconformer_amd import load_checkpoint, save_checkpoint
    cfg = named_config("Tiny")
    m = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, 3, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    ddp = {k.replace("encoder.", "encoder.module.", 1).replace("fc.", "fc.module.", 1): torch.from_numpy(v) for k, v in sd.items()}
    path = str(tmp_path / "ref.ckpt")
    torch.save({"model_state_dict": ddp, "optimizer_state_dict": {"state": {}}, "model_step": 1234, "tokenizer": {"fake": "tokenizer"},
                "is_distributed": True}, path)
    m.load(path)
    assert m.tokenizer == {"fake": "tokenizer"}
    assert torch.equal(m.fc.weight, torch.from_numpy(sd["fc.weight"]))
    assert torch.equal(m.encoder.blocks[2].norm.weight, torch.from_numpy(sd["encoder.blocks.2.norm.weight"]))
    out = str(tmp_path / "native.ckpt")
    save_checkpoint(m, out)
    m2 = ModelCTC.from_config(cfg)
    load_checkpoint(m2, out)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_collate_and_bucketing():
    """collate_fn_pad as the reference (utils/preprocessing.py:27-45); bucket plan deterministic, bounded, covering."""
    from efficientconformer_amd import bucket_batches, collate_fn_pad
    g = torch.Generator().manual_seed(0)
    lens = [5, 9, 3, 9, 7]
    batch = [[torch.randn(1, n, generator=g), torch.arange(1, 1 + n // 3)] for n in lens]
    data, target, dl, tl = collate_fn_pad(batch)
    assert dl.tolist() == [9, 9, 7, 5, 3] and data.shape == (5, 9) and tl.tolist() == [3, 3, 2, 1, 1]
    assert torch.equal(data[0, :9], batch[1][0].reshape(-1)) and torch.equal(data[1], batch[3][0].reshape(-1))   # stable for ties
    assert float(data[4, 3:].abs().sum()) == 0.0 and target.shape == (5, 3)
    d2, t2, dl2, tl2 = collate_fn_pad([b[0] for b in batch])
    assert t2 is None and tl2 is None and torch.equal(d2, data)
    lengths = synth.libri_lengths(300, seed=5).tolist()
    plan = bucket_batches(lengths, max_batch=64, max_padded_samples=64 * 160000)
    flat = [i for b in plan for i in b]
    assert sorted(flat) == list(range(300))
    for b in plan:
        assert len(b) <= 64 and len(b) * lengths[b[0]] <= 64 * 160000
        assert all(lengths[b[j]] >= lengths[b[j + 1]] for j in range(len(b) - 1))
    assert plan == bucket_batches(lengths, max_batch=64, max_padded_samples=64 * 160000)


def test_bench_batches_share_one_padded_length_across_ranks():
    """bench.py: the all-gather of encoder outputs needs one (B, T) shape on every rank."""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = types.SimpleNamespace(workload="libri", batch=3)       # tiny batches: per-rank maxima differ
    shapes = {bench.make_batch(args, r, 4)[0].shape for r in range(4)}
    assert len(shapes) == 1
    own = [int(synth.libri_lengths(3, seed=1234 + r).max()) for r in range(4)]
    assert len(set(own)) > 1 and max(own) == next(iter(shapes))[1]


def test_pmc_summary_feeds_bench_traffic_only_for_the_profiled_command(tmp_path, monkeypatch):
    """tools/pmc_summary.py turns two rocprofv3 --pmc databases into the text summary + profiles/pmc_traffic.json; bench.py's
    roofline.traffic must come from it ONLY when model, batch, workload AND --streams equal the profiled command's (launch shapes
    differ with the number of row ranges), with FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md)."""
    import json
    import sqlite3
    import subprocess
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "void (anonymous namespace)::chain_kernel<8, 8, 4, 1, false>((anonymous namespace)::ChainDev, unsigned long long*)"
    dbs = []
    for counter, vals in (("FETCH_SIZE", [1000.0, 3000.0]), ("WRITE_SIZE", [500.0, 700.0])):
        p = tmp_path / (counter + ".db")
        c = sqlite3.connect(p)
        c.execute("create table pmc_events (name text, counter_name text, counter_value real)")
        c.execute("create table kernels (name text, start integer, end integer)")
        for i, v in enumerate(vals):
            c.execute("insert into pmc_events values (?, ?, ?)", (name, counter, v))
            c.execute("insert into kernels values (?, ?, ?)", (name, 1000 * i, 1000 * i + 90000 + 4000 * i))
        c.commit(); c.close()
        dbs.append(str(p))
    txt, js = tmp_path / "pmc.txt", tmp_path / "pmc.json"
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), dbs[0], dbs[1], str(txt), "python bench.py", str(js),
                    "EfficientConformerCTCSmall", "256", "libri", "2"], check=True, capture_output=True)
    d = json.load(open(js))
    assert d["streams"] == 2 and d["batch"] == 256
    k = d["kernels"][name]
    assert k["calls"] == 2 and k["fetch_x2_bytes"] == 2 * 1024 * 2000.0 and k["write_bytes"] == 1024 * 600.0
    body = open(txt).read()
    assert "kernel durations" in body and "92.000" in body          # (90 + 94) / 2 us with dispatches serialised

    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.replace(js, tmp_path / "profiles" / "pmc_traffic.json")
    args = types.SimpleNamespace(model="EfficientConformerCTCSmall", batch=256, workload="libri", streams=2)
    pat = r"chain_kernel<\d+, \d+, \d+, [123],"
    assert bench.pmc_traffic(args, pat) == k["fetch_x2_bytes"] + k["write_bytes"]
    for other in (dict(streams=1), dict(batch=128), dict(model="EfficientConformerCTCMedium")):
        a2 = types.SimpleNamespace(**{**vars(args), **other})
        assert bench.pmc_traffic(a2, pat) is None


def _run_bench(argv, env_extra=None, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_flag_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` outside torchrun must start 2 ranks itself (the reference spawns one process per GPU:
    main.py:217-220) and rank 0 prints ONE JSON line with n_gpus = 2; --dry-run takes the same launcher with gloo and no GPU."""
    r, j = _run_bench(["--gpus", "2", "--dry-run", "--batch", "6"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1
    assert j["n_gpus"] == 2 and j["dry_run"] and j["config"]["global_batch"] == 12 and j["padded_samples_equal_on_all_ranks"]
    assert j["metric"] == "audio-frames/sec through encoder, EffConformerCTC-Small, 1/2/4/8 GPU"


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    r, j = _run_bench(["--gpus", "1", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_raw_ddp_state_dict_with_frontend_buffers_loads():
    """A raw DDP state dict carries `encoder.module.preprocessing.*` torchaudio buffers (model.py:367-370 strips `.module.`):
    the infix is stripped first, then the frontend buffers are dropped."""
    cfg = named_config("Tiny")
    m = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, 3, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    ddp = {k.replace("encoder.", "encoder.module.", 1): torch.from_numpy(v) for k, v in sd.items()}
    ddp["encoder.module.preprocessing.Spectrogram.window"] = torch.zeros(400)
    ddp["encoder.module.preprocessing.MelScale.fb"] = torch.zeros(257, 80)
    m.load_state_dict(ddp)                                   # strict
    assert torch.equal(m.encoder.linear.weight, torch.from_numpy(sd["encoder.linear.weight"]))


def test_no_product_kernel_carries_a_hazardous_packed_fp32_form():
    """ISA guard (tools/check_isa.py): every gfx950 code object of libeffconf.so is disassembled; no product kernel may contain
    v_pk_{add,mul,fma}_f32 with an op_sel low-lane swizzle (wrong results next to another wave's bf16 MFMA on MI355X:
    profiles/r2_mel_packed_fp32_hazard.txt).  The library is compiled with -target-feature -packed-fp32-ops."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(root, "tools", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _lib.load()
    rows = mod.scan(_lib.LIB_PATH)
    names = mod.demangle(list(rows))
    assert len(rows) > 100
    # round 5: the product library carries NO exempt kernel any more (csrc/debug.hip and the packed-fp32 mel build moved to libeffconf_debug.so):
    # every kernel of libeffconf.so is checked, none holds a packed-fp32 VALU instruction at all
    assert not [n for n in names.values() if mod.EXEMPT.search(n)]
    product = {names[k]: v for k, v in rows.items()}
    assert any("mel_kernel<0, 80>" in n for n in product) and any("chain_kernel" in n for n in product) and any("chain3_kernel" in n for n in product)
    assert all(v[0] == 0 and v[1] == 0 for v in product.values()), {n: v for n, v in product.items() if v[0] or v[1]}
    # round 6: the split-precision chain kernels hold no exec-masked region (a compiler-placed VGPR -> AGPR copy inside one corrupted a store address:
    # profiles/r6_35_side_bisect.txt); every other kernel may
    sxc = {n: v for n, v in product.items() if mod.BRANCH_FREE.search(n)}
    assert len(sxc) >= 20 and all(v[2] == 0 for v in sxc.values()), {n: v for n, v in sxc.items() if v[2]}
    # the diagnostic library: its mel build with packed fp32 (variant 8) is the one that carries the hazardous forms
    drows = mod.scan(os.path.join(os.path.dirname(_lib.LIB_PATH), "libeffconf_debug.so"))
    dnames = mod.demangle(list(drows))
    assert any(v[1] > 0 for k, v in drows.items() if "mel_pk_build" in dnames[k])


SHIPPED = ["ConformerCTCSmall", "ConformerCTCMedium", "ConformerCTCLarge", "ConformerTransducerSmall", "ConformerTransducerMedium",
           "ConformerTransducerLarge", "EfficientConformerCTCSmall", "EfficientConformerCTCMedium", "EfficientConformerCTCLarge",
           "EfficientConformerTransducerSmall", "EfficientConformerTransducerMedium", "EfficientConformerTransducerLarge"]


@pytest.mark.parametrize("name", SHIPPED)
def test_every_shipped_config_has_a_name_and_a_plan(name):
    """All 12 model configs the reference ships (configs/*.json) resolve by name; where the reference tree is present (build
    container) the named config must equal the JSON file key for key on everything the hot path reads."""
    import json
    cfg = named_config(name)
    plan = build_plan(cfg["encoder_params"])
    assert len(plan.blocks) == cfg["encoder_params"]["num_blocks"]
    path = os.path.join("/root/reference/configs", name + ".json")
    if not os.path.exists(path):
        return
    ref = json.load(open(path))
    ours, theirs = cfg["encoder_params"], ref["encoder_params"]
    for k in ("arch", "num_blocks", "dim_model", "ff_ratio", "num_heads", "kernel_size", "conv_stride", "att_stride", "strided_blocks", "expand_blocks",
              "att_group_size", "relative_pos_enc", "max_pos_encoding", "subsampling_module", "subsampling_layers", "subsampling_filters",
              "sample_rate", "win_length_ms", "hop_length_ms", "n_fft", "n_mels", "normalize", "mean", "std"):
        if k in theirs:
            assert ours.get(k) == theirs[k], (name, k, ours.get(k), theirs[k])
    assert cfg["tokenizer_params"]["vocab_size"] == ref["tokenizer_params"]["vocab_size"]
    for sect in ("decoder_params", "joint_params"):
        if sect in ref:
            for k, v in cfg[sect].items():
                assert ref[sect][k] == v, (name, sect, k)
    p2 = build_plan(theirs)
    assert [(b.dim_model, b.dim_expand, b.num_heads, b.kernel_size, b.group_size, b.conv_stride, b.max_pos) for b in plan.blocks] == \
           [(b.dim_model, b.dim_expand, b.num_heads, b.kernel_size, b.group_size, b.conv_stride, b.max_pos) for b in p2.blocks]


def test_host_pack_rows_is_the_reference_collate():
    """effconf_host_pack_rows (the front door's native packer) == pad_sequence of the reference's collate_fn_pad
    (utils/preprocessing.py:33-45) for every thread count, with and without zero-fill, incl. empty rows; bad arguments are errors."""
    import ctypes as C

    from efficientconformer_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    lens = [5000, 4096, 4095, 17, 1, 0, 0]
    rows = [torch.randn(n, generator=g) for n in lens]
    want = torch.nn.utils.rnn.pad_sequence(rows, batch_first=True, padding_value=0)
    n, pitch = len(rows), want.shape[1]
    ptrs = (C.c_void_p * n)(*[r.data_ptr() for r in rows])
    ln = torch.tensor(lens, dtype=torch.int64)
    lp = C.cast(ln.data_ptr(), C.POINTER(C.c_int64))
    for threads in (1, 2, 3, 8, 64):
        dst = torch.full((n, pitch), 7.0)
        assert lib.effconf_host_pack_rows(ptrs, lp, n, dst.data_ptr(), pitch, 1, threads) == 0
        assert torch.equal(dst, want)
        dst = torch.full((n, pitch), 7.0)
        assert lib.effconf_host_pack_rows(ptrs, lp, n, dst.data_ptr(), pitch, 0, threads) == 0
        for r in range(n):
            assert torch.equal(dst[r, :lens[r]], rows[r]) and bool((dst[r, lens[r]:] == 7.0).all())
    assert lib.effconf_host_pack_rows(ptrs, lp, n, dst.data_ptr(), 4999, 1, 2) != 0          # a row longer than the pitch
    assert lib.effconf_host_pack_rows(ptrs, lp, 0, None, 0, 1, 2) == 0
