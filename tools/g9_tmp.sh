cd /root/repo
for c in citation2 ppa; do echo "== $c"; AB_ARGS="--config $c" bash tools/ab_flags.sh "" "-DOCN_X_WALK_SET=2048 -DOCN_X_WALK_Q=512" "-DOCN_X_WALK_SET=1024 -DOCN_X_WALK_Q=512 -DOCN_X_WALK_BM_BITS=16"; done
