set -e
for caps in "3072 1024" "2048 1024" "4096 1536"; do
  set -- $caps
  echo "== caps closest $1 any $2"
  RT_WAVES_CLOSEST=$1 RT_WAVES_ANY=$2 timeout -k 10 100 python tools/ab/multi_overlap.py cfg4
done
