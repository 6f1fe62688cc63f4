import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/whisper.axera_amd/tools')
import whisper_axera_amd as wa, modelgen
mdir = '/tmp/axw_bench_models'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
e = wa.Whisper('small', mdir, 'zh', device=0, max_batch=B)
e.bench('encoder', B, 0, 2)
e.close()
