cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q5; mkdir -p $O
KICP_TRACE=1 timeout 120 python -c "
import numpy as np, kinematic_icp_amd as K
from kinematic_icp_amd import synthetic as syn
cfg, scene, scans, rng = syn.make_case('cfg1', n_scans=1)
m = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
syn.build_map_points(scene, cfg, m.AddPoints, m.num_points, rng)
r = K.KinematicRegistration()
s = scans[0]
print(r.ComputeRobotMotion(K.DeviceFrame(s['frame']), m, s['last_pose'], s['rel_odom'], cfg.first_frame_tau()), 'aql_active', r.get_option('aql_active'))
" 2>&1 | grep -v "ms$" | tail -5
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 300 python tools/gpu_blocks.py cfg2 256 2>&1 | tee $O/blocks_cfg2.txt
