"""Accuracy of the two depthwise-convolution kernels against the fp32 oracle, end to end (test infrastructure: imports oracle/)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from efficientconformer_amd import ModelCTC, named_config, synth
from oracle import ref_encoder as R

for name in sys.argv[1:] or ["ConformerCTCSmall", "EfficientConformerCTCSmall"]:
    cfg = named_config(name)
    m = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, 1, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    osd = {k[len("encoder."):] if k.startswith("encoder.") else k: v for k, v in sd.items()}
    m = m.cuda()
    rng = np.random.default_rng(7)
    lens = np.concatenate([np.array([3000, 3400, 4100, 5000, 6500, 8000]), (16000 * (0.6 + 2.4 * rng.random(10))).astype(np.int64)]).astype(np.int64)
    audio = torch.from_numpy(synth.make_audio(lens, seed=6)).cuda()
    ln = torch.from_numpy(lens).cuda()
    enc = m.encoder
    enc.ragged = False
    refs = []
    for b in range(len(lens)):
        li = int(lens[b])
        with torch.no_grad():
            ref, _ = R.encoder(audio[b:b + 1, :li].cpu(), ln[b:b + 1].cpu(), osd, enc.plan)
        refs.append(ref)
    for opt in (0, 1):
        enc.set_option("dwconv_mfma", opt)
        mx, mn, short_mx = [], [], []
        for b in range(len(lens)):
            li = int(lens[b])
            out, ol, _ = enc(audio[b:b + 1, :li].contiguous(), ln[b:b + 1].contiguous())
            d = (out.float().cpu() - refs[b]).abs()
            mx.append(float(d.max())); mn.append(float(d.mean()))
        print(name, "dwconv_mfma", opt, "max err per utterance: worst %.4f median %.4f | mean err: worst %.5f median %.5f | the six shortest (5 - 13 frames): %s"
              % (max(mx), float(np.median(mx)), max(mn), float(np.median(mn)), " ".join("%.4f" % v for v in mx[:6])))
