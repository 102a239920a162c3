# The reference's own dataset end to end through the headless harness: the JPEG as shipped (harness/jpeg_reader.hpp), the PNG annotation,
# 300 pipelined live frames each (annotation upload + estimate + map download per frame).  usage: bash scripts/dataset_live.sh
make -s -C harness
for f in tests/golden/dataset/*.jpg; do
    n=$(basename $f .jpg)
    out=$(harness/rtdd_harness -i $f -a tests/golden/dataset/${n}_ann.png -o /tmp/rtdd_${n}_ --live 300)
    echo "$n: $(echo "$out" | grep -o 'Processing Time: [0-9.]* ms') | $(echo "$out" | grep -o 'Live: [0-9.]* frames/s')"
done
