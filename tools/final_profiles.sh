#!/bin/bash
# Usage (GPU box, via gpurun): bash tools/final_profiles.sh <tag>
# The records of a finished tree in one call: the GPU test suite, smoke(), per-kernel MFMA-busy and HBM-side traffic of the fp16 pass (written
# into profiles/ on the box so that the bench line that follows carries them, and copied to gpurun_out/<tag>/), the default bench line, and the
# rocprofv3 --kernel-trace --stats summaries of the default and the fp16 command.  Copy what comes back under gpurun_out/<tag>/ into profiles/.
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log
bash tools/pmc.sh ${TAG}_busy "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" --no-sub-records --precision fp16 --batch 256 --steps 3 --warmup 1 > $O/pmc_busy.log 2>&1
python3 tools/mfma_busy.py gpurun_out/pmc_${TAG}_busy profiles/r04_f16_mfma_busy.json > $O/mfma_busy.txt && cp profiles/r04_f16_mfma_busy.json $O/
timeout -k 10 400 bash tools/f16_traffic.sh ${TAG} > $O/traffic.log 2>&1 && cp gpurun_out/f16_traffic_${TAG}.json profiles/r04_f16_traffic.json && cp gpurun_out/f16_traffic_${TAG}.json gpurun_out/f16_traffic_${TAG}_summary.txt $O/
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);f=d['fp16_b256'];print(d['value'],f['value'],f['whole_pass']['frac'],f['roofline']['traffic'],f['mfma_busy'].get('source'),d['fp32tol_b128']['value'],d['e2e_u8_b64']['value'])"
bash tools/profile.sh ${TAG}_i16 > /dev/null && cp gpurun_out/prof_${TAG}_i16_summary.txt $O/kernel_stats_int16.txt
bash tools/profile.sh ${TAG}_f16 --precision fp16 --batch 256 > /dev/null && cp gpurun_out/prof_${TAG}_f16_summary.txt $O/kernel_stats_fp16.txt
echo done
