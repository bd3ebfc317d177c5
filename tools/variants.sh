set -e
cd nsgp-repre_amd/csrc
for v in "-DNSGP_STAGGER=0 -DNSGP_ASSUME_ALIGNED=0" "-DNSGP_STAGGER=1 -DNSGP_ASSUME_ALIGNED=0" "-DNSGP_STAGGER=0 -DNSGP_ASSUME_ALIGNED=1"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $v -o ../libnsgp_repre_hip.so projected_step.hip covariance.hip projector.hip prototype.hip
  echo "== $v"
  (cd ../.. && timeout -k 10 200 python tools/plan_bench.py 2>&1 | grep -E "8x\(512,4096|R-50-FPN table  ")
done
