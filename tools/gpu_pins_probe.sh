cd $GRAFT_REPO_ROOT
for c in nn_34 nn_64; do
for pin in "" "ACME_COOP_GPW=2" "ACME_COOP_GPW=1" "ACME_COOP_GPW=1 ACME_COOP_WPB=4" "ACME_COOP_GPW=2 ACME_COOP_WPB=4" "ACME_COOP_GPW=1 ACME_COOP_IMGL=1" "ACME_COOP_GPW=2 ACME_COOP_IMGL=1" "ACME_COOP_GPW=4 ACME_COOP_IMGL=1"; do
echo "pins: $pin"; env $pin timeout 600 python tools/generic_shape_probe.py 8192 600 $c 2>&1 | tail -1
done; done
