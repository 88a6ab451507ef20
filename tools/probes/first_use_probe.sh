#!/bin/bash
for order in imshcnpPdke imhnpdk imkhnpd impPdk imkpPd imshcshc imhnhn imHn; do tools/probes/first_use_probe $order; done
