# Three harness live loops (three processes, three dataset pairs) on ONE GPU at the same time: frames/s each, what each healed, and that a shared
# run leaves the same map as a lone one.  usage: bash scripts/share_gpu.sh  (-> profiles/r06_shared_gpu.txt; round 5: r05_shared_gpu.txt)
make -s -C harness
H=harness/rtdd_harness
I=tests/golden/dataset
( $H -i $I/Rock.jpg -a $I/Rock_ann.png -o /tmp/a_ --live 3000 > /tmp/a.out 2>/tmp/a.err ) &
( $H -i $I/Pigs.jpg -a $I/Pigs_ann.png -o /tmp/b_ --live 3000 > /tmp/b.out 2>/tmp/b.err ) &
( $H -i $I/Dog.jpg -a $I/Dog_ann.png -o /tmp/c_ --live 3000 > /tmp/c.out 2>/tmp/c.err ) &
wait
for x in a b c; do echo "== $x"; grep -E "Live|Processing" /tmp/$x.out; sort /tmp/$x.err | uniq -c | cut -c1-260 | head -5; done
# and each result against a lone run's
$H -i $I/Rock.jpg -a $I/Rock_ann.png -o /tmp/a1_ --live 3000 > /tmp/a1.out 2>&1; cmp /tmp/a_DepthMap.pgm /tmp/a1_DepthMap.pgm && echo "Rock: shared == alone"
