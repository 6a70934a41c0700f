set -e
for t in 0 8 0 8; do
 GPEMU_GEMM_TABLE=$t timeout -k 10 300 python3 bench.py --no-cpu-baseline > /tmp/b_$t.json 2>/tmp/b_$t.err
 python3 -c "
import json,sys;b=json.loads(open('/tmp/b_$t.json').read().strip().splitlines()[-1]);print('table',$t,'value %.1f'%b['value'],'pred %.0f'%b['predictions']['value'],'frac %.4f'%b['roofline']['frac'],'us/launch %.1f'%b['roofline']['avg_launch_us'],'potrf %.4f'%b['roofline_other']['potrf_whole']['frac'],'predgemm %.3f'%b['roofline_other']['predict_gemm']['frac'])"
done
