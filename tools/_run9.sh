set -o pipefail
cd $GRAFT_REPO_ROOT
bash tools/_run7.sh
bash tools/_run8.sh
