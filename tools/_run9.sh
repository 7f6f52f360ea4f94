set -e
cd $GRAFT_REPO_ROOT
export STAMPS_MODE=defect
echo "=== B=1 reg32"; STAMPS_B=1 SCPN_HIP_TILEREG=1 SCPN_HIP_TILEREG_TINYB=0 timeout -k 10 100 python tools/tile_stamps.py
echo "=== B=1 reg24"; STAMPS_B=1 timeout -k 10 100 python tools/tile_stamps.py
echo "=== B=32 LDS tile"; STAMPS_B=32 SCPN_HIP_TILEREG=0 timeout -k 10 100 python tools/tile_stamps.py
echo "=== B=32 reg32"; STAMPS_B=32 SCPN_HIP_TILEREG=1 timeout -k 10 100 python tools/tile_stamps.py
echo "=== B=32 reg48"; STAMPS_B=32 SCPN_HIP_TILEREG=1 STAMPS_DEFINES="TILE_REG_LZ=48 TILE_REG_NW=12" timeout -k 10 100 python tools/tile_stamps.py
