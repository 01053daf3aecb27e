# the certificate kernel against the exact DP on ten times the instances of the test suite (2.7 M adversarial instances), then three more soak seeds
mkdir -p gpurun_out/r3
MP_THERMO_TEST_SCALE=10 timeout -k 10 900 python -m pytest tests/test_gpu_thermo.py -x -q -k never_rejects > gpurun_out/r3/thermo_campaign.log 2>&1 || { tail -20 gpurun_out/r3/thermo_campaign.log; exit 1; }
tail -2 gpurun_out/r3/thermo_campaign.log
