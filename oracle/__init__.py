"""CPU oracle for the avhot hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU restatement (NumPy, plus plain C under oracle/c/ for the pixel path) of
the reference's per-frame detect -> lane -> track -> KF -> plan algorithms
and of the two rule-based taggers that consume their outputs.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import anything from this package; the shipped package
(multimodal_autonomous_driving_perception_and_planning_amd/) never does and
fails loudly when its HIP library is missing.

Pinning status:
  detector_ref (D1), tracker_ref (K1-K3), planner_ref (P1-P3)
      pinned bit-for-bit / <=1e-12 against tests/golden/*.npz, which were
      produced by running the real reference classes (tests/golden/make_golden.py).
  maneuver_ref (T1), interaction_ref (T2)   -- the rule-based taggers of SURVEY section 8 f-3
      pinned bit-for-bit against maneuver.npz / interaction.npz / interaction_synth.npz, produced by
      the real src/tagging/maneuver_detector.py and interaction_detector.py (loaded from their files:
      NumPy only), the latter also behind the real detector -> tracker chain.
  kf_ref (E1-E3)      PARITY UNPINNED: filterpy is not installed and the
                      reference ships no tests; restates filterpy 1.4.5's
                      published predict/update equations.
  lane_ref (L1-L7)    PARITY UNPINNED: OpenCV is not installed; restates
                      OpenCV 4.x semantics for cvtColor/GaussianBlur/Canny/
                      fillPoly/HoughLinesP from their published algorithms.
  raster_ref (f-2)    PARITY UNPINNED: the reference draws with cv2; this states the
                      project's own pixel rules for the geometry of those calls.
  yolo_ref (D2)       PARITY UNPINNED: ultralytics is not installed and
                      yolov8n.pt is absent; PyTorch-CPU fp32 module with the
                      YOLOv8n topology and seeded random weights.

Reference citations are relative to /root/reference.
"""
