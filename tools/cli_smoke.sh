set -u
cd $(mktemp -d)
R=$GRAFT_REPO_ROOT
B="python $R/main.py --synthetic --vocab 300 --bs 4 --embed_dim 32 --enc_hid 64 --dec_hid 64 --latent 10 --gen_z_samples 4 --gpu 0 --epochs 1 --max_steps 3"
run() { echo "== $*"; "$@" > out.log 2>&1; rc=$?; tail -2 out.log; echo "rc=$rc"; }
run $B --prior GMM
run $B --no_encoder
run $B --optimizer SGD
run $B --optimizer Momentum --prior AG --c_v
run $B --prior AG --c_v --ann_param 2.0 --dec_drop 0.7 --dec_lstm_drop 0.8
run $B
run $B --restore
run $B --mode inference --sample_gen beam_search --gen_name bs
run $B --mode inference --sample_gen sample --temperature 0.7 --gen_name sm
run $B --fine_tune --bs 2 --save_params
run $B --fine_tune --bs 2 --mode inference --gen_name ft
ls
