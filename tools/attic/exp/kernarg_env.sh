R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python3 bench.py --no-cpu-baseline --no-extra --full-line --steps 20000 --warmup 2000 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 wall_us %6.3f stamped %6.3f' % (r['wall_us_per_launch'], r['kernel_us_stamped']))"; }
for rep in 1 2; do
unset HIP_FORCE_DEV_KERNARG; line "default               "
HIP_FORCE_DEV_KERNARG=1 line "HIP_FORCE_DEV_KERNARG=1"
HIP_FORCE_DEV_KERNARG=0 line "HIP_FORCE_DEV_KERNARG=0"
done
env | grep -i "^HIP_\|^HSA_\|^ROC" | head
