mkdir -p gpurun_out/r04
O=gpurun_out/r04/halves_caps.txt
: > $O
export HALVES_PARTS=1,2
echo "default caps" >> $O; timeout -k 10 120 python tools/ab/halves_time.py cfg4 8 >> $O 2>&1
export HALVES_PARTS=2
for caps in "4096 2048" "3072 1536" "2560 1280" "2048 1024" "5120 1024" "3072 3072"; do set -- $caps; echo "closest $1 any $2" >> $O; RT_WAVES_CLOSEST=$1 RT_WAVES_ANY=$2 timeout -k 10 120 python tools/ab/halves_time.py cfg4 8 >> $O 2>&1; done
cat $O
