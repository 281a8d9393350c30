# block shape sweep of the post-physics launch (GPU box): bash tools/sweep_post.sh "192:24 384:24 384:32 ..."
for spec in $1; do t=${spec%%:*}; e=${spec##*:}
echo "threads=$t envs=$e"; MDP_POST_THREADS=$t MDP_POST_ENVS=$e timeout -k 10 120 python3 tools/bench_post.py 2>/dev/null | tail -1 || exit 1
done
