COMMON="--dataset reddit --normalization graphsage --weight_decay 0 --dropout 0.2 --layer_norm --hidden1 128 --num_fc_layers 2 --epochs 4 --early_stopping 30 --batch_size=512 --test_batch_size=512 --cv --cvd --test_cv --degree=1 --test_degree=1"
for extra in "" "--prefetch 6" "--sampler_threads 2"; do
echo "== $extra"; timeout 600 python -m stochastic_gcn_amd.train $COMMON $extra 2>&1 | grep -E "sgcn\] epoch|Epoch" | sed -E 's/.*(time= [0-9.]+ ttime= [0-9.]+ \(sch [0-9.]+ s\)).*/\1/' | tail -6
done
