#!/bin/bash
scripts/probe/_probe_cu_stream_bw
