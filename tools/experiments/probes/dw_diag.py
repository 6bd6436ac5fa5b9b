"""Diagnostic: ragged vs alone with dwconv_mfma on / off (ConformerCTCSmall, kernel size 31)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from efficientconformer_amd import ModelCTC, named_config, synth

name = sys.argv[1] if len(sys.argv) > 1 else "ConformerCTCSmall"
cfg = named_config(name)
m = ModelCTC.from_config(cfg)
sd = synth.make_state_dict(m.encoder.plan, 3, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.cuda()
lens = np.array([70000, 52345, 33000, 20000, 8000], dtype=np.int64)
audio = torch.from_numpy(synth.make_audio(lens, seed=6)).cuda()
ln = torch.from_numpy(lens).cuda()
enc = m.encoder
for opt in (0, 1):
    enc.set_option("dwconv_mfma", opt)
    enc.ragged, enc.sub_batches, enc.trim_sub_batches = True, 1, False
    out, out_len, _ = enc(audio, ln, x_len_host=lens)
    enc.ragged = False
    for b in range(len(lens)):
        li = int(lens[b])
        alone, al, _ = enc(audio[b:b + 1, :li].contiguous(), ln[b:b + 1].contiguous())
        tb = int(al[0])
        d = (out[b, :tb].float() - alone[0].float()).abs()
        rows = torch.nonzero(d.amax(dim=1) > 0).flatten().tolist()
        print("dwconv_mfma", opt, "utt", b, "T", tb, "max", float(d.max()), "rows differing", len(rows), rows[:6], rows[-3:])
