COMMON="--dataset reddit --normalization graphsage --weight_decay 0 --dropout 0.2 --layer_norm --hidden1 128 --num_fc_layers 2 --batch_size=512 --test_batch_size=512 --epochs 5 --early_stopping 30"
for T in 8 16; do
echo "=== NS L=2 no PP degree 20, sampler_threads $T (native)"; timeout 600 python -m stochastic_gcn_amd.train $COMMON --degree=20 --test_degree=20 --nopreprocess --notest_preprocess --max_steps 120 --sampler_threads $T 2>&1 | grep -E "sgcn\] epoch" | cut -c1-100
echo "=== same, python threads"; timeout 600 python -m stochastic_gcn_amd.train $COMMON --degree=20 --test_degree=20 --nopreprocess --notest_preprocess --max_steps 120 --sampler_threads $T --nonative_prefetch 2>&1 | grep -E "sgcn\] epoch" | cut -c1-100
done
echo "=== CVD+PP"; timeout 600 python -m stochastic_gcn_amd.train $COMMON --cv --cvd --test_cv --degree=1 --test_degree=1 2>&1 | grep -E "sgcn\] epoch" | cut -c1-100
