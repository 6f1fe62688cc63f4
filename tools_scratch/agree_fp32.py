"""How often do the GPU's greedy ids equal the PURE-fp32 oracle's (not the bf16-policy one)?"""
import os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + '/oracle', R + '/whisper.axera_amd/tools', R + '/tests', R + '/tests/golden'):
    sys.path.insert(0, p)
import numpy as np, torch
import modelgen, oracle, whisper_axera_amd as wa
from conftest import load_demo_pcm
clips = [load_demo_pcm()] + [modelgen.synth_clip(i, 480000 if i % 2 else 150000) for i in range(1, 6)]
for mt, seed, n in (('micro', 11, 440), ('tiny', 14, 200), ('small', 0, 64)):
    with tempfile.TemporaryDirectory() as td:
        dims = modelgen.DIMS[mt]
        w = modelgen.synth_weights(dims, seed)
        modelgen.write_model_dir(td, mt, dims, weights=w)
        cfg = modelgen.make_config(mt, dims)
        e = wa.Whisper(mt, td, 'zh', device=0, max_batch=6)
        o32 = oracle.Oracle(cfg, w, bf16_policy=False, threads=32)
        got = e.run_tokens_batch(clips if mt != 'small' else clips[:2], max_new=n)
        tot = agree = 0
        for b, g in enumerate(got):
            mel, _, _ = oracle.log_mel(clips[b], dims['n_mels'])
            ck, cv = o32.encoder(mel)
            ids = o32.greedy(ck, cv, 'zh', max_new=n)
            k = 0
            while k < min(len(ids), len(g)) and ids[k] == g[k]:
                k += 1
            tot += len(ids); agree += k
            print(mt, 'clip', b, 'first divergence at', k, 'of', len(ids))
        print(mt, 'prefix agreement with the fp32 oracle: %d/%d' % (agree, tot))
        e.close()
