for t in 192 384 576 768; do for e in 16 24 32 64; do
echo "threads=$t envs=$e"; MDP_POST_THREADS=$t MDP_POST_ENVS=$e timeout -k 10 120 python tools/bench_post.py 2>/dev/null | tail -1 || exit 1
done; done
