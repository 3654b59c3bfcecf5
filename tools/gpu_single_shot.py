import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cwc_import; pkg = cwc_import.load()
import cwc_import
C = cwc_import.load().graphgen.circuits
data = C.build_authv2_class().to_bin()
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'circuit9_authV2_inputs.json')).read()
for i in range(int(os.environ.get('SHOTS', '14'))):
    t0 = time.perf_counter(); w = pkg.calc_witness_wtns(js, data); dt = time.perf_counter() - t0
    print("gw_calc_witness call %d: %.1f ms (%d bytes)" % (i, dt * 1e3, len(w)), flush=True)
    if i < 8:
        time.sleep(0.3)  # (the background search for the best program finishes meanwhile)
